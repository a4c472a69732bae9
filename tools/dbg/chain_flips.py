"""cfg4 bit decisions against the CPU oracle chain, per numeric mode: how many of a capture's ~250 k slicer decisions differ
(FAST is statistically tied, GENERIC bit-tied; round 3: FAST_REFTAPS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip(); po = grhip_loader.import_oracle(); wl = g.workload
c, c4 = wl.CFG2, wl.CFG4
S, n = int(os.environ.get("S", "6")), 10_000_000
nout = n // 4
dev = torch.device("cuda", 0); st = torch.cuda.Stream(device=dev)
xs = [wl.fsk4_capture(n, stream_id=500 + s) for s in range(S)]
d_in = torch.empty((S, n, 2), dtype=torch.float32, device=dev)
for s in range(S):
    d_in[s] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
refs = []
for s in range(S):
    dem = po.chain_xlating_demod(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"], xs[s])
    sym, _ = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
    refs.append(po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(sym)))
ch = g.dmr_chain(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"], c4["omega"], c4["gain_omega"], c4["mu"],
                 c4["gain_mu"], c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
d_bits = torch.zeros((S, nout), dtype=torch.uint8, device=dev); d_n = torch.zeros(S, dtype=torch.int32, device=dev)
for name, mode in (("FAST", g.MODE_FAST), ("FAST_REFTAPS", g.MODE_FAST_REFTAPS), ("FAST_VALU", g.MODE_FAST_VALU), ("GENERIC", g.MODE_GENERIC)):
    ch.set_mode(mode)
    d_bits.zero_(); torch.cuda.synchronize()
    ch.run_device(d_in, n, n, d_bits, nout, d_n, st); st.synchronize()
    nb = d_n.cpu().numpy(); bits = d_bits.cpu().numpy()
    flips, lens = [], []
    for s in range(S):
        same = int(nb[s]) == len(refs[s])
        lens.append(same)
        flips.append(int(np.count_nonzero((bits[s, :nb[s]] ^ refs[s][:nb[s]]) & 1)) if same else -1)
    print("%-13s symbol counts equal: %s; slicer decisions that differ from the oracle's, per capture of %d symbols: %s; flags equal: %s"
          % (name, all(lens), len(refs[0]), flips, all(int((bits[s, :nb[s]] >> 1).sum()) == int((refs[s] >> 1).sum()) for s in range(S))), flush=True)
