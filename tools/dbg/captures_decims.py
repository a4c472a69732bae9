"""run_captures (64 captures, one batched launch) at decimations only the direct kernel takes: rate"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip()
wl = g.workload
c = wl.CFG2
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
S, n = 64, 10_000_000
x = wl.fsk4_capture(n, stream_id=3)
xt = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
d_in = torch.empty((S, n, 2), dtype=torch.float32, device=dev)
for s in range(S):
    d_in[s] = xt
for ntaps, decim in ((400, 20), (200, 10), (100, 5), (256, 8), (320, 16), (96, 3)):
    nout = n // decim
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    blk = g.xlating_demod(decim, proto, c["center_freq"], c["fs"], 2.0)
    d_out = torch.empty((S, nout), dtype=torch.float32, device=dev)
    for _ in range(3):
        blk.run_captures_device(S, n, d_in, n, d_out, nout, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        blk.run_captures_device(S, n, d_in, n, d_out, nout, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%4d taps D=%2d: 64 captures of 10 M in %.3f ms = %.1f Gsamples/s (%.2f of the HBM peak)" % (
        ntaps, decim, ms, S * n / ms / 1e6, (S * n * 8 + S * nout * 4) / ms / 1e6 / 8000), flush=True)
