"""Diagnostic: per-phase time shares of the matrix-core FIR kernel (stamp build).
Run on the GPU box:  python tools/stamp_report_mfma.py   (uses libgrhip_stamp.so from `make stamp`)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GRHIP_LIB", os.path.join(ROOT, "gnuradio-3.5.0-dmr_amd", "libgrhip_stamp.so"))
import torch
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
c = wl.CFG2
dev = torch.device("cuda", 0)
B, n = 16, 10_000_000
x = wl.fsk4_capture(n)
buf = torch.zeros((B, n, 2), dtype=torch.float32, device=dev)
buf[:] = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
nout = n // 4
out = torch.empty((B, nout), dtype=torch.float32, device=dev)
blk = g.xlating_demod(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"])
st = torch.cuda.Stream(device=dev)
nwaves = 512 * 4
stamps = torch.zeros((nwaves, 10), dtype=torch.int64, device=dev)
L = g.lib()
L.grdbg_set_stamp_buffer_mfma.argtypes = [C.c_void_p]
assert L.grdbg_set_stamp_buffer_mfma(C.c_void_p(stamps.data_ptr())) == 0
for _ in range(40):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(st)
        blk.run_captures_device(B, n, buf, n, out, nout, st)
        e1.record(st)
st.synchronize()
print("kernel (events): %.1f us" % (e0.elapsed_time(e1) * 1e3))
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, :8].sum(1) > 0]
t0 = s[:, 8].min()
print("wave start after first start: mean %.1f us, max %.1f us;  wave end: min %.1f mean %.1f max %.1f us"
      % ((s[:, 8] - t0).mean() / 100, (s[:, 8] - t0).max() / 100, (s[:, 9] - t0).min() / 100,
         (s[:, 9] - t0).mean() / 100, (s[:, 9] - t0).max() / 100))
names = ["max+slot", "barrierA", "stage", "barrierB", "matrix(+fetch issue)", "epilogue", "-", "loop-top"]
tiles_total = B * ((nout + 1983) // 1984)
tot = s[:, :8].sum(1)
print("waves reporting:", len(s), " mean per wave: %.1f us; tiles per WG %.1f" % (tot.mean() / 100.0, tiles_total / 512.0))
for k, nme in enumerate(names):
    print("  %-22s %6.2f %%   (%.3f us / tile)" % (nme, 100 * s[:, k].mean() / tot.mean(), s[:, k].mean() / (tiles_total / 512.0) / 100.0))
# per wave-in-workgroup shares (does one wave lag?)
full = stamps.cpu().numpy().astype(np.float64).reshape(-1, 4, 10)
for wv in range(4):
    sh = full[:, wv, :8].mean(0)
    print("wave %d:" % wv, " ".join("%5.1f" % (100 * v / sh.sum()) for v in sh))
