# chain at 2048 captures: number of time slices (same box, interleaved)
mkdir -p gpurun_out; rm -f gpurun_out/chain_chunks.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do for v in diag pc48 pc64; do
  echo "== $v" >> gpurun_out/chain_chunks.log
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 300 python tools/bench_chain.py 2048 10000000 2>/dev/null | tail -1 | cut -c150-330 >> gpurun_out/chain_chunks.log || exit 1
done; done
cat gpurun_out/chain_chunks.log
