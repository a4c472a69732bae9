"""gr_framer_sink_1 on one GPU: 64 M correlator items, a 100-byte packet about every
PERIOD items (default 1000).  usage: python tools/bench_framer.py [PERIOD ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import grhip_loader

g = grhip_loader.import_grhip()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
rng = np.random.default_rng(0)


def timeit(fn, reps=10):
    fn(); st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


n = 64_000_000
hb = np.array([(((5 << 12) | 100) >> (15 - i)) & 1 for i in range(16)] * 2, dtype=np.uint8)
for period in [int(a) for a in sys.argv[1:]] or [1000]:
    xb = rng.integers(0, 2, n, dtype=np.uint8)
    starts = np.arange(100, n - 2000, period)
    for k in range(32):
        xb[starts + k] = hb[k]
    xb[starts] |= 2
    db = torch.from_numpy(xb).to(dev)
    fs = g.framer_sink_1()
    ms = timeit(lambda: fs.work_device(n, db, st))
    fs.messages(st)
    t0 = time.perf_counter()
    fs.work_device(n, db, st)
    got = fs.messages(st)
    dt = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"block": "framer_sink_1", "items": n, "packet_every": period, "messages_per_call": len(got),
                      "kernels_ms": round(ms, 3), "kernels_Mbits_per_s": round(n / ms / 1e3, 1),
                      "with_fetch_ms": round(dt, 2), "with_fetch_Mbits_per_s": round(n / dt / 1e3, 1)}), flush=True)
