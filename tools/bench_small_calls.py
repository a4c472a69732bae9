"""Per-call cost of the host-buffer block API at scheduler-sized calls (what a GNU Radio flowgraph would see)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
c = wl.CFG2
proto = wl.cfg2_proto_taps()
x = wl.fsk4_capture(4_000_000)
xin = wl.with_history(x, len(proto) - 1)


def per_call(blk, nout, decim, item_in=1):
    n_calls = min(200, (len(x) // decim) // nout)
    # warm
    for k in range(3):
        blk.work(nout, xin[k * nout * decim: (k + 1) * nout * decim + len(proto) - 1])
    t0 = time.perf_counter()
    for k in range(n_calls):
        blk.work(nout, xin[k * nout * decim: (k + 1) * nout * decim + len(proto) - 1])
    return (time.perf_counter() - t0) / n_calls


for nout in (1024, 8192, 65536):
    fz = g.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    t = per_call(fz, nout, 4)
    xl = g.freq_xlating_fir_filter_ccc(c["decim"], proto, c["center_freq"], c["fs"])
    t2 = per_call(xl, nout, 4)
    print(json.dumps({"noutput_items": nout, "xlating_demod.work us": round(t * 1e6, 1),
                      "Msamples/s": round(nout * 4 / t / 1e6, 1),
                      "freq_xlating.work us": round(t2 * 1e6, 1), "xl Msamples/s": round(nout * 4 / t2 / 1e6, 1)}))
