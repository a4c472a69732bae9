import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import grhip_loader
g = grhip_loader.import_grhip(); po = grhip_loader.import_oracle(); wl = g.workload
c = wl.CFG2
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = wl.fsk4_capture(n, stream_id=3)
proto = wl.cfg2_proto_taps()
nout = n // 4
ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
blk = g.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
got = blk.work(nout, wl.with_history(x, 255))
e = np.abs(got - ref) / np.abs(ref).max()
bad = np.nonzero(e > 1e-5)[0]
print("n_out", nout, "bad count", len(bad), "first", bad[:20], "mod 2016:", (bad[:20] % 2016), "max err", e.max())
for i in bad[:6]:
    print(i, got[i], ref[i])
