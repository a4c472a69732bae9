import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import grhip_loader
g = grhip_loader.import_grhip(); po = grhip_loader.import_oracle()
rng = np.random.default_rng(0)
bad = 0
for nt in range(1, 33):
    for ol in range(1, 18):
        x = np.rint(rng.uniform(-1, 1, nt + ol - 1 + 4) * 32768).astype(np.float32)
        t = np.rint(rng.uniform(-1, 1, nt) * 32768).astype(np.float32)
        ref = np.array([np.dot(x[o:o + nt].astype(np.float64), t[::-1].astype(np.float64)) for o in range(ol)])
        blk = g.fir_filter_fff(1, t)
        got = blk.filterNdec(x, ol, 1)
        e = np.abs(got - ref) > np.abs(ref) * 9e-3
        if e.any():
            bad += 1
            if bad < 6: print("nt", nt, "ol", ol, "bad at", np.nonzero(e)[0], got[e][:3], ref[e][:3])
print("bad cases", bad)
