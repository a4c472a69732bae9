import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip()
po = grhip_loader.import_oracle()
wl = g.workload
c, c4 = wl.CFG2, wl.CFG4
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
for decim, ntaps, n_out, S, extra, pad in ((20, 400, 70_000, 2, 0, 0), (20, 400, 70_000, 3, 0, 0), (20, 400, 70_000, 2, 19, 0), (20, 400, 70_000, 2, 0, 33),
                                    (20, 400, 70_000, 3, 19, 33)):
    n = n_out * decim + extra
    stride = n + pad
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    xs = [wl.fsk4_capture(n, stream_id=50 + s) for s in range(S)]
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    for omega in (c4["omega"] * 4 / decim,):
        ch = g.dmr_chain(decim, proto, c["center_freq"], c["fs"], c["demod_gain"], omega, c4["gain_omega"], c4["mu"],
                         c4["gain_mu"], c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
        for mode in (g.MODE_FAST, g.MODE_GENERIC, g.MODE_FAST_VALU):
            ch.set_mode(mode)
            ch.run_device(d_in, n, stride, d_bits, n_out, d_n, st)
            st.synchronize()
            nb = d_n.cpu().numpy()
            p_dem, s_dem = ch.intermediate(0)
            p_soft, s_soft = ch.intermediate(1)
            for s in range(S):
                dem = np.empty(n_out, np.float32)
                g.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * s * s_dem), n_out * 4)
                soft_mine, _ = po.chain_mm(omega, c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
                soft = np.empty(nb[s], np.float32)
                g.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft), int(nb[s]) * 4)
                m = min(len(soft), len(soft_mine))
                bad = np.nonzero(soft[:m].view(np.uint32) != soft_mine[:m].view(np.uint32))[0]
                print("S=%d extra=%d pad=%d " % (S, extra, pad) + "D=%d n_out=%d omega=%.2f mode=%d s=%d: n %d vs %d, mismatches %d first %s" % (decim, n_out, omega, mode, s, len(soft), len(soft_mine), len(bad), bad[:8]), flush=True)
