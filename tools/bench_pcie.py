"""PCIe-inclusive rate of the host-buffer boundary: grhip_xlating_demod_work() on numpy
arrays (pageable host memory, H2D + kernel + D2H + synchronise per call), one capture."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process)
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
c = wl.CFG2
n = 10_000_000
x = wl.fsk4_capture(n)
proto = wl.cfg2_proto_taps()
xin = wl.with_history(x, len(proto) - 1)
nout = n // c["decim"]
blk = g.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
for _ in range(3):
    blk.reset(); blk.work(nout, xin)
ts = []
for _ in range(10):
    blk.reset()
    t0 = time.perf_counter(); blk.work(nout, xin); ts.append(time.perf_counter() - t0)
t = float(np.median(ts))
print(json.dumps({"what": "xlating_demod.work() on host buffers, one 10 M-sample capture per call (pageable memory)",
                  "ms_per_call": t * 1e3, "Msamples_per_s": n / t / 1e6,
                  "host_bytes_moved_GBps": (8 * len(xin) + 4 * nout) / t / 1e9}))
