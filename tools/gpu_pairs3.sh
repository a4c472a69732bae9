# mm_pairs_kernel alone (64 captures, 32 per wave): timing-only ablations
mkdir -p gpurun_out; rm -f gpurun_out/pairs_abl.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for v in diag mmp3 mmp2 mmp4; do
  echo "== $v" >> gpurun_out/pairs_abl.log
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 200 python tools/bench_chain.py 64 10000000 --cpw 32 2>/dev/null | tail -1 | cut -c150-330 >> gpurun_out/pairs_abl.log || exit 1
done
cat gpurun_out/pairs_abl.log
