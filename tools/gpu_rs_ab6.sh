# Round 3: role-split kernel v5 (epilogue split over the halves): tests, stamps, A/B; PFB hier + FFT filter tests
mkdir -p gpurun_out; rm -f gpurun_out/rs_ab10.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 400 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu > gpurun_out/rs_tests.log 2>&1
rc=$?; tail -5 gpurun_out/rs_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/stamp_report_rs.py > gpurun_out/rs_stamps.log 2>&1; cat gpurun_out/rs_stamps.log
for rep in 1 2; do
for v in ${VARIANTS:-diag:0 diag:1}; do
  lib=${v%%:*}; rs=${v##*:}
  env GRHIP_LIB=$L/libgrhip_$lib.so GRHIP_MF_RS=$rs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']),'frac',round(d['roofline']['frac'],4))" >> gpurun_out/rs_ab10.log || exit 1
done; done
cat gpurun_out/rs_ab10.log
timeout -k 10 900 python -m pytest tests/test_gpu_fft_pfb.py -x -q -m gpu -k "hier" > gpurun_out/pfb_tests.log 2>&1; tail -6 gpurun_out/pfb_tests.log
