# bit-exact window kernel, whole batch (bench line's bit_exact_engine): tap operand scalar / copied to vector registers
mkdir -p gpurun_out; rm -f gpurun_out/generic_ab2.log
L=$PWD/gnuradio-3.5.0-dmr_amd
for rep in 1 2; do for v in diag gwv; do
  GRHIP_LIB=$L/libgrhip_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --chain-captures 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); b = d['bit_exact_engine']
print('$v', 'generic kernel_ms', round(b['kernel_ms'], 4), 'Gsamples/s', round(b['Msamples_per_s_per_gpu'] / 1e3, 1), '| FAST', round(d['roofline']['kernel_ms'], 4))" >> gpurun_out/generic_ab2.log || exit 1
done; done
cat gpurun_out/generic_ab2.log
