"""the fused xlating -> demodulator block at decimations other than 2 / 4: rate and parity (FAST mode)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip()
po = grhip_loader.import_oracle()
wl = g.workload
import parity_util
c = wl.CFG2
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
for ntaps, decim in ((400, 20), (200, 10), (100, 5), (64, 8), (320, 16)):
    n = 40_000_000 // decim * decim
    x = wl.fsk4_capture(n, stream_id=3)
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    nout = n // decim
    ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], 2.0, x)
    blk = g.xlating_demod(decim, proto, c["center_freq"], c["fs"], 2.0)
    got = g.run_sync_block(blk, x, chunk=700_000)
    ok, worst = parity_util.demod_close(got, ref, skip=max(64, ntaps // decim + 1), gain=2.0)
    xin = torch.from_numpy(wl.with_history(x, ntaps - 1).view(np.float32).reshape(-1, 2)).to(dev)
    y = torch.empty(nout, device=dev)
    blk2 = g.xlating_demod(decim, proto, c["center_freq"], c["fs"], 2.0)
    for _ in range(5):
        blk2.reset(); blk2.work_device(nout, xin, y, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        blk2.reset(); blk2.work_device(nout, xin, y, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 10
    # streaming: the same buffer fed again and again WITHOUT reset (the rotator phase keeps advancing)
    import time
    blk3 = g.xlating_demod(decim, proto, c["center_freq"], c["fs"], 2.0)
    blk3.work_device(nout, xin, y, st); st.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        blk3.work_device(nout, xin, y, st)
    st.synchronize()
    ms_s = (time.perf_counter() - t0) / 5 * 1e3
    print("%4d taps D=%2d: parity %s (%s)  %.1f Gsamples/s from a fresh handle, %.1f streaming" % (ntaps, decim, ok, worst if not ok else "", n / ms / 1e6, n / ms_s / 1e6), flush=True)
