mkdir -p gpurun_out
L=$PWD/gnuradio-3.5.0-dmr_amd
GRHIP_LIB=$L/libgrhip_diag.so timeout -k 10 900 python -m pytest tests/test_gpu_fir_mfma.py tests/test_gpu_fir.py -x -q > gpurun_out/spf_tests.log 2>&1; rc=$?; tail -2 gpurun_out/spf_tests.log
[ $rc -eq 0 ] || exit $rc
VARIANTS="GRHIP_LIB=$L/libgrhip_diag.so GRHIP_LIB=$L/libgrhip_spf0.so" bash tools/gpu_ab.sh > gpurun_out/spf_ab.log 2>&1; cat gpurun_out/spf_ab.log
