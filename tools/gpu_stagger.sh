mkdir -p gpurun_out; rm -f gpurun_out/stagger.log
for sg in 0 4000 8000 12000 16000 24000; do
  GRHIP_STAGGER=$sg timeout -k 10 200 python bench.py --steps 10 --warmup 3 --captures 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stagger',$sg,'kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/stagger.log
done
cat gpurun_out/stagger.log
timeout -k 10 300 python -m pytest tests/test_gpu_fir.py tests/test_gpu_chain.py tests/test_gpu_host_cpp.py -m gpu -q 2>&1 | tail -3
