"""stress parity of the fused xlating->demod batched path: S captures of n samples, every output compared with the oracle"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import grhip_loader
from parity_util import demod_close
g = grhip_loader.import_grhip(); po = grhip_loader.import_oracle(); wl = g.workload
c = wl.CFG2
S, n = 6, 2_000_000
proto = wl.cfg2_proto_taps()
nout = n // 4
dev = torch.device("cuda", 0)
d_in = torch.zeros((S, n, 2), dtype=torch.float32, device=dev)
xs = [wl.fsk4_capture(n, stream_id=40 + s) for s in range(S)]
for s in range(S):
    d_in[s] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
d_out = torch.zeros((S, nout), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(device=dev)
blk = g.xlating_demod(4, proto, c["center_freq"], c["fs"], c["demod_gain"])
for rep in range(3):
    blk.run_captures_device(S, n, d_in, n, d_out, nout, st)
st.synchronize()
got = d_out.cpu().numpy()
worst = 0.0; nbad = 0
for s in range(S):
    ref = po.chain_xlating_demod(4, proto, c["center_freq"], c["fs"], c["demod_gain"], xs[s])
    ok, w = demod_close(got[s], ref, gain=c["demod_gain"])
    err = np.abs(got[s][64:] - ref[64:]); bad = np.nonzero(err > 1e-5 * np.abs(ref).max())[0]
    nbad += len(bad); worst = max(worst, w)
    if not ok: print("stream", s, "FAIL", w, bad[:10] + 64)
print("stress: %d streams x %d outputs, worst steady rel err %.3g, samples over 1e-5: %d" % (S, nout, worst, nbad))
