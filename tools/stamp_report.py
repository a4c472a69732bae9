"""Diagnostic: per-phase cycle shares of the tiled FIR kernel (stamp build).
Run on the GPU box:  GRHIP_LIB=.../libgrhip_stamp.so python tools/stamp_report.py"""
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
B, n = 8, 10_000_000
x = wl.fsk4_capture(n)
buf = torch.zeros((B, n, 2), dtype=torch.float32, device=dev)
buf[:] = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
nout = n // 4
out = torch.empty((B, nout), dtype=torch.float32, device=dev)
blk = g.xlating_demod(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"])
st = torch.cuda.Stream(device=dev)
nwaves = 512 * 4
stamps = torch.zeros((nwaves, 8), dtype=torch.int64, device=dev)
L = g.lib()
L.grdbg_set_stamp_buffer.argtypes = [C.c_void_p]
assert L.grdbg_set_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
for _ in range(3):
    blk.run_captures_device(B, n, buf, n, out, nout, st)
st.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
s = s[s.sum(1) > 0]
names = ["stage", "barrier1", "fetch-issue", "predecessor", "MAC", "epilogue", "barrier2", "loop-top"]
tot = s.sum(1)
print("waves reporting:", len(s), " mean cycles per wave (one launch): %.0f" % tot.mean())
for k, nme in enumerate(names):
    print("  %-12s %6.2f %%   (%.0f cycles / tile)" % (nme, 100 * s[:, k].mean() / tot.mean(), s[:, k].mean() / (9768 / 512)))
