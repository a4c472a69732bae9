# Round 3: role-split kernel v4 (chunk-major matrix phase): tests, stamps, A/B; generic tiled kernel tests + rate; chain tests
mkdir -p gpurun_out; rm -f gpurun_out/rs_ab5.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 400 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu > gpurun_out/rs_tests.log 2>&1
rc=$?; tail -5 gpurun_out/rs_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/stamp_report_rs.py > gpurun_out/rs_stamps.log 2>&1; cat gpurun_out/rs_stamps.log
for rep in 1 2; do
for v in ${VARIANTS:-diag:0 diag:1}; do
  lib=${v%%:*}; rs=${v##*:}
  env GRHIP_LIB=$L/libgrhip_$lib.so GRHIP_MF_RS=$rs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']),'frac',round(d['roofline']['frac'],4))" >> gpurun_out/rs_ab5.log || exit 1
done; done
cat gpurun_out/rs_ab5.log
timeout -k 10 900 python -m pytest tests/test_gpu_fir.py -x -q -m gpu > gpurun_out/fir_tests.log 2>&1; tail -4 gpurun_out/fir_tests.log
timeout -k 10 300 python tools/dbg/generic_rate.py > gpurun_out/generic_rate.log 2>&1; tail -6 gpurun_out/generic_rate.log
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "output_limit or full_size" > gpurun_out/chain_new_tests.log 2>&1; tail -5 gpurun_out/chain_new_tests.log
