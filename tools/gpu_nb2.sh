# shipped FIR kernel with tiles of half the size (GRHIP_MF_NBLK=2), two / three workgroups per CU: parity on the D = 4 shapes, then A/B
mkdir -p gpurun_out; rm -f gpurun_out/nb2_tests.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for v in nb2 nb2w3; do
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_fir.py -x -q -k "cfg2 or batched_vs_oracle" >> gpurun_out/nb2_tests.log 2>&1; echo "$v tests rc=$?" | tee -a gpurun_out/nb2_tests.log
done
tail -3 gpurun_out/nb2_tests.log
VARIANTS="GRHIP_LIB=$L/libgrhip_diag.so GRHIP_LIB=$L/libgrhip_nb2.so GRHIP_LIB=$L/libgrhip_nb2w3.so" bash tools/gpu_ab.sh > gpurun_out/nb2_ab.log 2>&1; cat gpurun_out/nb2_ab.log
