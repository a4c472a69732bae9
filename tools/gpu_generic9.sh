# bit-exact window kernel, whole batch: lanes per workgroup 256 (shipped) / 128 / 64
mkdir -p gpurun_out; rm -f gpurun_out/generic_ab3.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do for v in diag gw128 gw64; do
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --chain-captures 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); b = d['bit_exact_engine']
print('$v', 'generic kernel_ms', round(b['kernel_ms'], 4), 'Gsamples/s', round(b['Msamples_per_s_per_gpu'] / 1e3, 1), '| FAST', round(d['roofline']['kernel_ms'], 4))" >> gpurun_out/generic_ab3.log || exit 1
done; done
cat gpurun_out/generic_ab3.log
GRHIP_LIB=$L/libgrhip_gw128.so timeout -k 10 600 python -m pytest tests/test_gpu_fir.py -x -q -k "generic or bit_exact" 2>&1 | tail -2
