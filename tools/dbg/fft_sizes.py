"""gr_fft_vcc by size: rate and fraction of the HBM peak (16 B per sample)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import grhip_loader
g = grhip_loader.import_grhip()
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
tot = 1 << 26
for N in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
    nvec = tot // N
    x = torch.randn((tot, 2), device=dev); y = torch.empty((tot, 2), device=dev)
    f = g.fft_vcc(N, True, [], False)
    for _ in range(30): f.work_device(nvec, x, y, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): f.work_device(nvec, x, y, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("N=%5d  %7.1f Gsamples/s  frac %.3f" % (N, tot / ms / 1e6, tot * 16 / (ms * 1e-3) / 8e12), flush=True)
