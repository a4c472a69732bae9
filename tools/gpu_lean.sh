# shipped FIR kernel: lean demodulator in the epilogue (diag) against quad_demod_fast (lean0); parity first
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fir_mfma.py tests/test_gpu_fir.py tests/test_gpu_chain.py -x -q > gpurun_out/lean_tests.log 2>&1; rc=$?; tail -3 gpurun_out/lean_tests.log; grep "cfg2 demod parity" gpurun_out/lean_tests.log | head -3
[ $rc -eq 0 ] || exit $rc
L=$PWD/gnuradio-3.5.0-dmr_amd
VARIANTS="GRHIP_LIB=$L/libgrhip_diag.so GRHIP_LIB=$L/libgrhip_lean0.so" bash tools/gpu_ab.sh > gpurun_out/lean_ab.log 2>&1; cat gpurun_out/lean_ab.log
