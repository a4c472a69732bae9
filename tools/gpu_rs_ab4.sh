# Round 3: role-split kernel v3: tests, stamps, A/B against the round-2 kernel (same box, interleaved), chain tests, full bench line
mkdir -p gpurun_out; rm -f gpurun_out/rs_ab4.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 400 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu > gpurun_out/rs_tests.log 2>&1
rc=$?; tail -5 gpurun_out/rs_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/stamp_report_rs.py > gpurun_out/rs_stamps.log 2>&1; cat gpurun_out/rs_stamps.log
for rep in 1 2; do
for v in ${VARIANTS:-diag:0 diag:1}; do
  lib=${v%%:*}; rs=${v##*:}
  env GRHIP_LIB=$L/libgrhip_$lib.so GRHIP_MF_RS=$rs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']),'frac',round(d['roofline']['frac'],4))" >> gpurun_out/rs_ab4.log || exit 1
done; done
cat gpurun_out/rs_ab4.log
timeout -k 10 500 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "output_limit or full_size" > gpurun_out/chain_new_tests.log 2>&1; tail -5 gpurun_out/chain_new_tests.log
timeout -k 10 500 python bench.py > gpurun_out/r03_bench_try.json 2> gpurun_out/r03_bench_try.err; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_try.json')); print('value',d['value'],'frac',d['roofline']['frac']); print(json.dumps(d.get('chain'))[:1500])"
