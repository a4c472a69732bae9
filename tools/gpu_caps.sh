# headline bench at several batch sizes (captures per launch)
mkdir -p gpurun_out; rm -f gpurun_out/caps.log
for c in ${CAPS:-2 4 8 16 32 64}; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --captures $c --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('captures',$c,'kernel_ms',round(d['roofline']['kernel_ms'],5),'ms_per_capture',round(d['roofline']['kernel_ms']/$c,5),'value',round(d['value']))" >> gpurun_out/caps.log
done
cat gpurun_out/caps.log
