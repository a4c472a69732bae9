import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import grhip_loader
g = grhip_loader.import_grhip(); po = grhip_loader.import_oracle(); wl = g.workload
ntaps, decim, n = 256, 4, 7001
x = wl.fsk4_capture(n * decim, stream_id=9)
proto = wl.lowpass_taps(ntaps, 200e3, 10e6).astype(np.complex64)
xin = wl.with_history(x, ntaps - 1)
ref = po.Xlating(decim, proto, 1.25e6, 10e6).work(xin, n)
blk = g.freq_xlating_fir_filter_ccc(decim, proto, 1.25e6, 10e6)
got = blk.work(n, xin)
err = np.abs(got - ref)
bad = np.nonzero(err > 1e-5 * np.abs(ref).max())[0]
print("nbad", len(bad), bad[:40], bad[-10:] if len(bad) else "")
for k in bad[:8]:
    print(k, got[k], ref[k], got[k] / ref[k])
