mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q -k "generic or bit_exact or cfg2 or chain_matches or mode_switch" > gpurun_out/generic_tests.log 2>&1; rc=$?; tail -5 gpurun_out/generic_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/dbg/generic_rate.py 2>/dev/null > gpurun_out/generic_rate.log; cat gpurun_out/generic_rate.log
timeout -k 10 300 python tools/dbg/generic_rate.py 100000000 2>/dev/null > gpurun_out/generic_rate100.log; cat gpurun_out/generic_rate100.log
