"""Diagnostic: per-role, per-phase time shares of fir_mfma_rs_kernel (stamp build: `make stamp`).
Run on the GPU box:  python tools/stamp_report_rs.py"""
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
NW = 12
nwaves = 256 * NW
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
full = stamps.cpu().numpy().astype(np.float64).reshape(-1, NW, 10)
tiles_total = B * ((nout + 1983) // 1984)
per_wg = tiles_total / 256.0
print("tiles per workgroup %.1f; periods = tiles + 3" % per_wg)
names_s = ["accumulator tiles -> registers", "epilogue (one block)", "convert + plane stores + next loads", "barrier A", "-", "barrier B", "wait for the tile + maxima", "loop"]
names_m = ["first chunks", "-", "-", "barrier A", "chunks + accumulators + two blocks of the epilogue", "barrier B", "-", "loop"]
for role, waves, names in (("stager", range(0, 8), names_s), ("matrix", range(8, 12), names_m)):
    s = full[:, list(waves), :].reshape(-1, 10)
    s = s[s[:, :8].sum(1) > 0]
    tot = s[:, :8].sum(1).mean()
    print("%s waves: %d reporting, mean lifetime %.1f us = %.3f us per period" % (role, len(s), tot / 100, tot / 100 / (per_wg + 3)))
    for k, nme in enumerate(names):
        if nme != "-":
            print("   %-28s %6.2f %%   (%.3f us / period)" % (nme, 100 * s[:, k].mean() / tot, s[:, k].mean() / (per_wg + 3) / 100.0))
for wv in range(NW):
    sh = full[:, wv, :8].mean(0)
    print("wave %2d:" % wv, " ".join("%5.1f" % (100 * v / max(sh.sum(), 1)) for v in sh))
