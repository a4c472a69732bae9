# 32 captures per wave: the loop alone (64 captures: the FIR is small beside it), each form; the chain at 2048 under rocprofv3;
# the 4FSK tail in slices: its tests, then the chain at 2048
mkdir -p gpurun_out; rm -f gpurun_out/pairs_alone.log
for cpw in 1 8 32; do
  timeout -k 10 200 python tools/bench_chain.py 64 10000000 --cpw $cpw 2>/dev/null | tail -1 | cut -c150-330 >> gpurun_out/pairs_alone.log || exit 1
done
cat gpurun_out/pairs_alone.log
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_digital.py -x -q -k "four_level or thirty_two or eight_captures" > gpurun_out/tail_tests.log 2>&1; rc=$?; tail -5 gpurun_out/tail_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_chain_prof.sh pairs_chain2048 2048 10000000 --cpw 32
timeout -k 10 300 python tools/bench_chain.py 2048 10000000 --four 2>/dev/null | tail -1 | cut -c150-330
