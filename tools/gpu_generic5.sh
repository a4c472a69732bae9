# bit-exact window kernel: where the taps come from (same box, interleaved)
mkdir -p gpurun_out; rm -f gpurun_out/generic_ab.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do for v in diag gwnopf gwlds; do
  echo "== $v" >> gpurun_out/generic_ab.log
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 200 python tools/dbg/generic_rate.py 2>/dev/null | head -1 >> gpurun_out/generic_ab.log || exit 1
done; done
cat gpurun_out/generic_ab.log
GRHIP_LIB=$L/libgrhip_gwlds.so timeout -k 10 600 python -m pytest tests/test_gpu_fir.py -x -q -k "generic or bit_exact" 2>&1 | tail -2
