mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --chain-captures 0 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; rc=$?
python3 -c "
import json
d=json.loads(open('gpurun_out/bench_quick.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac']); print(d.get('bit_exact_engine')); print(d.get('vector_engine',{}).get('Msamples_per_s_per_gpu'))
"
exit $rc
