mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q > gpurun_out/generic8_tests.log 2>&1; rc=$?; tail -4 gpurun_out/generic8_tests.log
[ $rc -eq 0 ] || exit $rc
# the whole chain in the bit-exact mode, 256 and 1024 captures
timeout -k 10 400 python tools/bench_chain.py 256 10000000 --generic 2>/dev/null | tail -1 | cut -c1-420
timeout -k 10 400 python tools/bench_chain.py 1024 10000000 --generic 2>/dev/null | tail -1 | cut -c1-420
