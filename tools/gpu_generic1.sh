# bit-exact mode: window kernel parity, then its rate (and the LDS-operand kernel's, same box)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q -k "generic or bit_exact or four_level or thirty_two" > gpurun_out/generic_tests.log 2>&1; rc=$?; tail -5 gpurun_out/generic_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/dbg/generic_rate.py 2>/dev/null > gpurun_out/generic_rate.log; cat gpurun_out/generic_rate.log
GRHIP_GENERIC_NO_WINDOW=1 timeout -k 10 300 python tools/dbg/generic_rate.py 2>/dev/null > gpurun_out/generic_rate_nowin.log; cat gpurun_out/generic_rate_nowin.log
