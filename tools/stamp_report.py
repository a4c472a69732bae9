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
stamps = torch.zeros((nwaves, 10), dtype=torch.int64, device=dev)
L = g.lib()
L.grdbg_set_stamp_buffer.argtypes = [C.c_void_p]
assert L.grdbg_set_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
for _ in range(3):
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
s = s[:, :8]
names = ["stage", "barrier1", "fetch-issue", "predecessor", "MAC", "epilogue", "barrier2", "loop-top"]
tot = s.sum(1)
print("waves reporting:", len(s), " mean ticks per wave (one launch, 100 MHz): %.0f = %.1f us" % (tot.mean(), tot.mean() / 100.0))
for k, nme in enumerate(names):
    print("  %-12s %6.2f %%   (%.2f us / tile)" % (nme, 100 * s[:, k].mean() / tot.mean(), s[:, k].mean() / (9768 / 512) / 100.0))

# ---- where does the spread between workgroups come from? ----
full = stamps.cpu().numpy().astype(np.float64)
dur = (full[:, 9] - full[:, 8]).reshape(-1, 4).max(1) / 100.0        # per workgroup, us
G = len(dur)
ids = np.arange(G)
tiles_total = B * ((nout + 2047) // 2048)
ntile = tiles_total // G + (ids < tiles_total % G)
print("workgroups %d, tiles per WG %d..%d" % (G, ntile.min(), ntile.max()))
print("per-tile time by WG: mean %.2f us  min %.2f  max %.2f  (p10 %.2f p90 %.2f)" % (
    (dur / ntile).mean(), (dur / ntile).min(), (dur / ntile).max(),
    np.quantile(dur / ntile, 0.1), np.quantile(dur / ntile, 0.9)))
pt = dur / ntile
for name, key in (("blockIdx % 8 (XCD?)", ids % 8), ("blockIdx // 256 (first/second WG of a CU?)", ids // 256),
                  ("(blockIdx // 8) % 4", (ids // 8) % 4)):
    print(name, " ".join("%.2f" % pt[key == k].mean() for k in np.unique(key)))
order = np.argsort(pt)
print("slowest WGs:", order[-12:], "fastest:", order[:12])
