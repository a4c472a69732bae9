# A/B of the headline bench under environment knobs / library variants, same box, interleaved.
# VARIANTS: space-separated list of "ENV=VALUE" settings (use X=1 for the default build).
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do
for v in ${VARIANTS:-X=1 GRHIP_NO_DIRECT=1}; do
  v2=${v//@L@/$L}
  env $v2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/ab.log
done; done
cat gpurun_out/ab.log
