# clock recovery, 32 captures per wave: parity tests, then the chain at 2048 captures with 8 / 32 captures per wave
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py -x -q -k "thirty_two or eight_captures or output_limit" > gpurun_out/pairs_tests.log 2>&1; rc=$?; tail -15 gpurun_out/pairs_tests.log
[ $rc -eq 0 ] || exit $rc
for cpw in 8 32 8 32; do
  timeout -k 10 300 python tools/bench_chain.py 2048 10000000 --cpw $cpw 2>/dev/null | tail -1 | cut -c1-420 >> gpurun_out/pairs_chain.log || exit 1
done
cat gpurun_out/pairs_chain.log
