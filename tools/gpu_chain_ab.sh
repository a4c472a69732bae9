# A/B of library variants on the full chain, same box (run on the GPU box): usage: bash tools/gpu_chain_ab.sh "S args" lib1 lib2 ...
L=$PWD/gnuradio-3.5.0-dmr_amd
args=$1; shift
for v in "$@"; do
  echo -n "== $v [$args]: " >> gpurun_out/chain_ab.log
  GRHIP_LIB=$L/$v timeout -k 10 200 python tools/bench_chain.py $args 2>/dev/null | grep samples_per_stream | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_batch'],2),'ms',round(d['Msamples_per_s']/1e3,1),'GS/s')" >> gpurun_out/chain_ab.log || exit 1
done
