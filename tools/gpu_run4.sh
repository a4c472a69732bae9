mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/t4.log 2>&1; echo "pytest exit $?" >> gpurun_out/t4.log; tail -12 gpurun_out/t4.log
rm -f gpurun_out/ablate.log
for ab in 0 1 2 4 3 7; do
  GRHIP_ABLATE=$ab timeout -k 10 200 python bench.py --steps 10 --warmup 3 --captures 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ablate',$ab,'kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
