mkdir -p gpurun_out; rm -f gpurun_out/caps2.log
run() { timeout -k 10 300 python bench.py --steps $3 --warmup 3 --captures $1 --samples $2 --no-cpu-baseline --chain-captures 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('captures',$1,'samples',$2,'steps',$3,'kernel_ms',round(d['roofline']['kernel_ms'],5),'value',round(d['value']))" >> gpurun_out/caps2.log; }
run 16 10000000 10
run 16 10000000 100
run 16 40000000 10
run 64 10000000 10
run 64 2500000 10
run 4 160000000 10
run 256 2500000 10
cat gpurun_out/caps2.log
