# quick GPU check: FIR/chain parity suites, then the headline bench (no CPU baseline) under ablations
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py -m gpu -q -x > gpurun_out/q.log 2>&1; echo "pytest exit $?" >> gpurun_out/q.log; tail -8 gpurun_out/q.log
rm -f gpurun_out/ablate.log
for ab in ${ABLATES:-0 4}; do
  GRHIP_ABLATE=$ab timeout -k 10 200 python bench.py --steps 10 --warmup 3 --captures 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ablate',$ab,'kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
