L=$PWD/gnuradio-3.5.0-dmr_amd
rm -f gpurun_out/chain_ab.log
for S in 1024 1536; do
for v in libgrhip.so libgrhip_p16.so libgrhip_p16c0.so libgrhip_p8c0.so; do
  echo "== $v S=$S" >> gpurun_out/chain_ab.log
  GRHIP_LIB=$L/$v timeout -k 10 200 python tools/bench_chain.py $S 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_batch'],2),'ms',round(d['Msamples_per_s']/1e3,1),'GS/s')" >> gpurun_out/chain_ab.log
done; done
cat gpurun_out/chain_ab.log
