# shipped FIR kernel: staging two rounds at a time in a hand-ordered block (diag) against the compiler's order (st0); parity first
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir_mfma.py tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q > gpurun_out/stage2_tests.log 2>&1; rc=$?; tail -3 gpurun_out/stage2_tests.log; grep "cfg2 demod parity" gpurun_out/stage2_tests.log | head -2
[ $rc -eq 0 ] || exit $rc
L=$PWD/gnuradio-3.5.0-dmr_amd
VARIANTS="GRHIP_LIB=$L/libgrhip_diag.so GRHIP_LIB=$L/libgrhip_st0.so" bash tools/gpu_ab.sh > gpurun_out/stage2_ab.log 2>&1; cat gpurun_out/stage2_ab.log
