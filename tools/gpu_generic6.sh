mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q > gpurun_out/generic6_tests.log 2>&1; rc=$?; tail -5 gpurun_out/generic6_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/dbg/generic_rate.py 2>/dev/null | head -1
