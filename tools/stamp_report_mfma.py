"""Diagnostic: per-phase time shares of fir_mfma_kernel, the shipped matrix-core FIR (stamp build: `make stamp`).
Run on the GPU box:  python tools/stamp_report_mfma.py"""
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
B, n = 64, 10_000_000
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
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    blk.run_captures_device(B, n, buf, n, out, nout, st)
    e1.record(st)
st.synchronize()
print("kernel (events): %.1f us" % (e0.elapsed_time(e1) * 1e3))
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, :8].sum(1) > 0]
life = (s[:, 9] - s[:, 8])
acc = s[:, :8]
names = ["tile maxima (block floating point)", "barrier (planes free)", "scale / pre-mix / split / plane stores", "barrier (planes written)",
         "matrix phase + interleaved epilogue", "last block's epilogue + carries", "-", "loop top / tile bookkeeping"]
print("waves reporting: %d, mean lifetime %.1f us (100 MHz ticks)" % (len(s), life.mean() / 100))
for k, nme in enumerate(names):
    print("  %-42s %6.2f %%" % (nme, 100 * acc[:, k].mean() / life.mean()))
print("  %-42s %6.2f %%" % ("(unstamped)", 100 * (1 - acc.sum(1).mean() / life.mean())))
