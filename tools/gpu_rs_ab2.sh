# Round 3: phase stamps of the role-split kernel, variants A/B (same box, interleaved), any-size FFT tests.
mkdir -p gpurun_out; rm -f gpurun_out/rs_ab2.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 200 python tools/stamp_report_rs.py > gpurun_out/rs_stamps.log 2>&1; cat gpurun_out/rs_stamps.log
for rep in 1 2; do
for v in ${VARIANTS:-diag:0 diag:1 pd4:1 prio1:1 prio2:1}; do
  lib=${v%%:*}; rs=${v##*:}
  env GRHIP_LIB=$L/libgrhip_$lib.so GRHIP_MF_RS=$rs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']),'frac',round(d['roofline']['frac'],4))" >> gpurun_out/rs_ab2.log || exit 1
done; done
cat gpurun_out/rs_ab2.log
timeout -k 10 600 python -m pytest tests/test_gpu_fft_pfb.py -x -q -m gpu -k "any_size or errors" > gpurun_out/fft_any_tests.log 2>&1; tail -15 gpurun_out/fft_any_tests.log
