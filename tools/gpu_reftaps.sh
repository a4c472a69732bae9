mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_fir_mfma.py -x -q -m gpu 2>&1 | tail -2
for m in 3 0 3 0; do python - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, grhip_loader
g = grhip_loader.import_grhip(); wl = g.workload; c = wl.CFG2
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev)
B, n = 64, 10_000_000
x = torch.randn((B, n, 2), device=dev); nout = n // 4; y = torch.empty((B, nout), device=dev)
blk = g.xlating_demod(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"]); blk.set_mode($m)
for _ in range(300): blk.run_captures_device(B, n, x, n, y, nout, st)
st.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(20): blk.run_captures_device(B, n, x, n, y, nout, st)
e1.record(st); st.synchronize()
ms = e0.elapsed_time(e1) / 20
print("mode $m: %.4f ms per 64 x 10 M samples = %.0f Gsamples/s, frac %.3f" % (ms, B * n / ms / 1e6, 9 * B * n / (ms * 1e-3) / 8e12))
PY
done
GRHIP_MODE=fast_reftaps timeout -k 10 300 python tools/dbg/demod_attrib.py 2>&1 | grep "GPU FAST (matrix"
