"""gr_pfb_channelizer_ccf oversampled by an integer factor (round 3: os launches of the fast kernel, one per residue of the
output index) and the hier block in one call: rates, and fractions of the HBM peak at the algorithmic bytes of each shape
(8 B in per input sample + 8 os B out; the hier form at os = 1: 16 B per sample)"""
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
tot = 1 << 25          # input samples (all channels together)
for M, tpf, osr in ((8, 32, 2), (8, 32, 4), (16, 16, 2), (4, 32, 4), (6, 20, 2), (6, 20, 1.5), (12, 16, 3)):
    taps = wl.lowpass_taps(M * tpf, 0.5 / M, 1.0)
    pf = g.pfb_channelizer_ccf(M, taps, float(osr))
    tc = tot // M                                   # items per stream
    nout = int(tc * osr)
    nout -= nout % pf.output_multiple()
    per = tc + tpf + 8
    xs = torch.randn((M * per, 2), device=dev); yo = torch.empty((nout * M, 2), device=dev)
    pf.general_work_device(nout, xs, per, yo, st)
    for _ in range(10): pf.general_work_device(nout, xs, per, yo, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): pf.general_work_device(nout, xs, per, yo, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = tot * 8 + nout * M * 8
    print("M=%3d taps/filter %3d oversample %4.1f  %7.1f Gsamples/s of input  frac %.3f" % (M, tpf, osr, tot / ms / 1e6, byt / (ms * 1e-3) / 8e12), flush=True)
    del xs, yo
tot = 1 << 26
for M, tpf in ((8, 32), (4, 32), (16, 16), (2, 32)):
    taps = wl.lowpass_taps(M * tpf, 0.5 / M, 1.0)
    pf = g.pfb_channelizer_ccf(M, taps, 1.0)
    nout = (tot // M) // 512 * 512
    xil = torch.randn(((nout + tpf) * M, 2), device=dev)
    yo = torch.empty((M, nout, 2), device=dev)
    pf.hier_work_device(nout, xil, yo, nout, st)
    for _ in range(10): pf.hier_work_device(nout, xil, yo, nout, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): pf.hier_work_device(nout, xil, yo, nout, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("hier block, one call: M=%3d taps/filter %3d  %7.1f Gsamples/s  frac %.3f (16 B per sample)" % (M, tpf, nout * M / ms / 1e6, nout * M * 16 / (ms * 1e-3) / 8e12), flush=True)
    del xil, yo
