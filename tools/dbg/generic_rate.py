"""rate of the bit-exact (GENERIC) path of the headline shape, one capture, device resident"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
proto = wl.cfg2_proto_taps()
x = torch.randn((n + 256, 2), device=dev)
nout = n // 4
y = torch.empty(nout, device=dev)
for mode in ("MODE_GENERIC", "MODE_FAST_VALU", "MODE_FAST"):
    blk = g.xlating_demod(4, proto, c["center_freq"], c["fs"], c["demod_gain"])
    blk.set_mode(getattr(g, mode))
    for _ in range(3):
        blk.reset(); blk.work_device(nout, x, y, st)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        blk.reset(); blk.work_device(nout, x, y, st)
    e1.record(st); st.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-15s %8.3f ms per %d M samples = %7.1f Gsamples/s" % (mode, ms, n // 1000000, n / ms / 1e6), flush=True)
