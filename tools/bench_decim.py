"""Decimating filters at shapes the tiled kernel does not take: which engine wins where.
usage: python tools/bench_decim.py   (GRHIP_NO_HIDEC=1 forces the overlap-save engine)"""
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
wl = g.workload
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)


def timeit(fn, reps=10, ramp_s=0.3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(5):
            fn()
        st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


n = 160_000_000
c = wl.CFG2
x = torch.randn((n + 4096, 2), device=dev)
for ntaps, D in ((400, 20), (400, 16), (100, 5), (200, 10), (64, 8), (1000, 25), (2000, 50), (40, 10), (128, 3)):
    y = torch.empty((n // D, 2), device=dev)
    cproto = (wl.lowpass_taps(ntaps, 0.4 / D, 1.0) * np.exp(0.01j * np.arange(ntaps))).astype(np.complex64)
    blk = g.freq_xlating_fir_filter_ccc(D, cproto, c["center_freq"], c["fs"])
    def run():
        blk.reset(); blk.work_device(n // D, x, y, st)
    ms = timeit(run)
    blkr = g.freq_xlating_fir_filter_ccc(D, wl.lowpass_taps(ntaps, 0.4 / D, 1.0).astype(np.complex64), c["center_freq"], c["fs"])
    def runr():
        blkr.reset(); blkr.work_device(n // D, x, y, st)
    msr = timeit(runr)
    blk2 = g.fir_filter_ccf(D, wl.lowpass_taps(ntaps, 0.4 / D, 1.0))
    ms2 = timeit(lambda: blk2.work_device(n // D, x, y, st))
    print(json.dumps({"ntaps": ntaps, "decim": D, "xlating_complex_proto_Msps": round(n / ms / 1e3, 1),
                      "xlating_real_proto_Msps": round(n / msr / 1e3, 1),
                      "fir_ccf_Msps": round(n / ms2 / 1e3, 1)}), flush=True)
