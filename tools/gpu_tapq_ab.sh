mkdir -p gpurun_out; rm -f gpurun_out/tapq_ab.log
L=$PWD/gnuradio-3.5.0-dmr_amd
timeout -k 10 600 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu > gpurun_out/rs_tests.log 2>&1; tail -3 gpurun_out/rs_tests.log
for rep in 1 2 3; do
for v in notapq diag; do
  env GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --captures 64 --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v','kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4))" >> gpurun_out/tapq_ab.log || exit 1
done; done
cat gpurun_out/tapq_ab.log
timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep "GPU FAST (matrix"
