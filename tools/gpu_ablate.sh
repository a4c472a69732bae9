# usage: bash tools/gpu_ablate.sh [pytest-args...]   (runs on the GPU box)
mkdir -p gpurun_out
if [ -n "$1" ]; then
  timeout -k 10 600 python -m pytest "$@" -m gpu -q > gpurun_out/t.log 2>&1; echo "pytest exit $?" >> gpurun_out/t.log; tail -8 gpurun_out/t.log
fi
rm -f gpurun_out/ablate.log
for ab in 0 1 2 4 3 7; do
  GRHIP_ABLATE=$ab timeout -k 10 200 python bench.py --steps 10 --warmup 3 --captures 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ablate',$ab,'kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/ablate.log
done
cat gpurun_out/ablate.log
