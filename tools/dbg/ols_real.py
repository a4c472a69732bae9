"""where does the real-data overlap-save engine differ from the direct form?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import grhip_loader
g = grhip_loader.import_grhip()
po = grhip_loader.import_oracle()
for ntaps, decim, n in [(1500, 1, 20001), (1500, 1, 4_000_000), (500, 8, 700_000)]:
    rng = np.random.default_rng(1)
    nin = n * decim + ntaps - 1
    x = rng.uniform(-1, 1, nin).astype(np.float32)
    taps = (rng.uniform(-1, 1, ntaps) / ntaps ** 0.5).astype(np.float32)
    blk = g.fir_filter_fff(decim, taps); blk.set_mode(g.MODE_FAST)
    got = blk.work(n, x)
    ref = po.fir_fff(taps, x, n, decim)
    err = np.abs(got - ref)
    bad = np.nonzero(err > 1e-5 * np.abs(ref).max())[0]
    print(ntaps, decim, n, "bad:", len(bad), bad[:10], bad[-10:] if len(bad) else "", "max", err.max())
    if len(bad):
        d = np.diff(bad); cuts = np.nonzero(d > 1)[0]
        print("  runs:", [(int(bad[0 if i == 0 else cuts[i-1]+1]), int(bad[c])) for i, c in enumerate(list(cuts[:8]) + [len(bad)-1])][:9])
