"""BASELINE configs[3] blocks alone (fft_vcc 4096, pfb_channelizer_ccf M=8): the cfg3 rows of
tools/bench_blocks.py.  usage: python tools/bench_cfg3.py"""
import json
import time
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
PEAK = 8000.0


def timeit(fn, reps=50, warm=3, ramp_s=0.3):
    # an idle MI355X needs tens of ms of load before its shader clock reaches steady state
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(10):
            fn()
        st.synchronize()
    for _ in range(warm):
        fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, ms, items, bytes_per_item, unit="Msamples/s"):
    gbs = items * bytes_per_item / (ms * 1e-3) / 1e9
    print(json.dumps({"block": name, "ms": round(ms, 4), "rate": round(items / ms / 1e3, 1), "unit": unit,
                      "algorithmic_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK, 4)}), flush=True)


rng = np.random.default_rng(0)

# cfg3: fft_vcc 4096-pt and pfb_channelizer M=8 (256-tap prototype) over 2^24 samples (268 MB of traffic: about the size
# of the part's last-level cache, a 60 us kernel) and over 2^27 (2.1 GB: steady state against HBM)
sizes = [int(a) for a in sys.argv[1:]] or [24, 27]
for lg in sizes:
    N, nvec = 4096, (1 << lg) // 4096
    xv = torch.randn((N * nvec, 2), device=dev)
    yv = torch.empty((N * nvec, 2), device=dev)
    ff = g.fft_vcc(N, True, [], False)
    report("fft_vcc 4096-pt x %d (2^%d samples)" % (nvec, lg), timeit(lambda: ff.work_device(nvec, xv, yv, st)), N * nvec, 16)
    del xv, yv
    M, nout = 8, (1 << lg) // 8
    taps = wl.lowpass_taps(256, 0.5 / M, 1.0)
    pf = g.pfb_channelizer_ccf(M, taps, 1.0)
    per = nout + 64
    xs = torch.randn((M * per, 2), device=dev)
    yo = torch.empty((nout * M, 2), device=dev)
    pf.general_work_device(nout, xs, per, yo, st)       # first call returns 0 (d_updated)
    report("pfb_channelizer_ccf M=8 256t (2^%d samples)" % lg, timeit(lambda: pf.general_work_device(nout, xs, per, yo, st)), nout * M, 16)
    del xs, yo

