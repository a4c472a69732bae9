"""per-call time of the fused block at scheduler-sized calls, by engine"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import grhip_loader
g = grhip_loader.import_grhip()
wl = g.workload
c = wl.CFG2
proto = wl.cfg2_proto_taps()
x = wl.fsk4_capture(4_000_000)
xin = wl.with_history(x, len(proto) - 1)
for nout in (256, 1024, 4096, 8192, 16384, 65536):
    row = {"nout": nout}
    for name in ("MODE_FAST", "MODE_FAST_VALU", "MODE_GENERIC"):
        blk = g.xlating_demod(4, proto, c["center_freq"], c["fs"], c["demod_gain"])
        blk.set_mode(getattr(g, name))
        n_calls = min(200, (len(x) // 4) // nout)
        for k in range(3):
            blk.work(nout, xin[k * nout * 4: (k + 1) * nout * 4 + len(proto) - 1])
        t0 = time.perf_counter()
        for k in range(n_calls):
            blk.work(nout, xin[k * nout * 4: (k + 1) * nout * 4 + len(proto) - 1])
        row[name] = round((time.perf_counter() - t0) / n_calls * 1e6, 1)
    print(row, flush=True)
