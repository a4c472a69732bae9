"""gr_pfb_channelizer_ccf by channel count and filter length (oversample 1): rate and fraction of the HBM peak"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip()
wl = g.workload
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
tot = 1 << 26
for M, tpf in ((2, 32), (4, 32), (8, 32), (16, 32), (8, 64), (8, 16), (32, 16), (32, 32), (64, 16), (128, 16), (5, 20), (10, 20), (12, 32), (15, 16)):
    nout = (tot // M) // 512 * 512
    taps = wl.lowpass_taps(M * tpf, 0.5 / M, 1.0)
    pf = g.pfb_channelizer_ccf(M, taps, 1.0)
    per = nout + 128
    xs = torch.randn((M * per, 2), device=dev); yo = torch.empty((nout * M, 2), device=dev)
    pf.general_work_device(nout, xs, per, yo, st)
    for _ in range(20): pf.general_work_device(nout, xs, per, yo, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): pf.general_work_device(nout, xs, per, yo, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("M=%3d taps/filter %3d  %7.1f Gsamples/s  frac %.3f" % (M, tpf, tot / ms / 1e6, tot * 16 / (ms * 1e-3) / 8e12), flush=True)
    del xs, yo
