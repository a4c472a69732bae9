"""gr_pfb_decimator_ccf on one GPU, device resident.  usage: python tools/bench_pfbdec.py"""
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


for M, ntaps in ((8, 256), (8, 64), (16, 512), (4, 128), (32, 256)):
    n = (1 << 27) // M                                        # 2^27 input samples in all
    tpf = -(-ntaps // M)
    per = n + tpf
    x = torch.randn((M, per, 2), device=dev)
    y = torch.empty((n, 2), device=dev)
    blk = g.pfb_decimator_ccf(M, wl.lowpass_taps(ntaps, 0.4 / M, 1.0), 1)
    blk.work_device(n, x, per, y, st)
    ms = timeit(lambda: blk.work_device(n, x, per, y, st))
    nin = n * M
    gbs = (nin * 8 + n * 8) / (ms * 1e-3) / 1e9
    print(json.dumps({"block": "pfb_decimator_ccf", "decim": M, "ntaps": ntaps, "ms": round(ms, 4),
                      "input_Msamples_per_s": round(nin / ms / 1e3, 1), "algorithmic_GBps": round(gbs, 1),
                      "frac_of_hbm_peak": round(gbs / 8000.0, 4),
                      "TFLOPs": round(nin * tpf * 4 / (ms * 1e-3) / 1e12, 1)}), flush=True)
