# Round 3: parity of the role-split matrix-core FIR kernel, then the headline bench A/B against the
# round-2 kernel on the same box (diagnostic build, GRHIP_MF_RS selects the kernel), interleaved.
mkdir -p gpurun_out; rm -f gpurun_out/rs_ab.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 400 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu > gpurun_out/rs_tests.log 2>&1
rc=$?; tail -5 gpurun_out/rs_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
for v in ${VARIANTS:-0 1}; do
  env GRHIP_LIB=$L/libgrhip_diag.so GRHIP_MF_RS=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures ${CAPTURES:-64} --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('GRHIP_MF_RS=$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']),'frac',round(d['roofline']['frac'],4), d['roofline'].get('kernel'))" >> gpurun_out/rs_ab.log || exit 1
done; done
cat gpurun_out/rs_ab.log
timeout -k 10 200 python -m pytest tests/test_gpu_fir.py -q -m gpu -k "parity_numbers" -s > gpurun_out/rs_parity.log 2>&1; grep -i "per element\|steady\|passed\|failed" gpurun_out/rs_parity.log | tail -5
timeout -k 10 300 python tools/dbg/demod_attrib.py > gpurun_out/rs_attrib.log 2>&1; cat gpurun_out/rs_attrib.log
