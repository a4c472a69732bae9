#!/usr/bin/env python3
"""Clock-recovery forms side by side on one batch (debugging aid): the chain with 1 / 8 / 32 captures per wave, first symbol that
differs per capture.  usage: python tools/dbg/mm_forms.py [decim ntaps n_out omega S mode]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import grhip_loader

g = grhip_loader.import_grhip()
wl = g.workload
a = sys.argv[1:]
decim, ntaps, n_out, omega, S = int(a[0]), int(a[1]), int(a[2]), float(a[3]), int(a[4])
mode = {"fast": g.MODE_FAST, "generic": g.MODE_GENERIC}[a[5]] if len(a) > 5 else g.MODE_FAST
c, c4 = wl.CFG2, wl.CFG4
n = n_out * decim + 1
proto = wl.cfg2_proto_taps() if decim == 4 else wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
gain = c["demod_gain"] * 4 / decim
dev = torch.device("cuda", 0)
stride = n + 7
d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
for s in range(S):
    d_in[s, :n] = torch.from_numpy(wl.fsk4_capture(n, stream_id=60 + s).view(np.float32).reshape(-1, 2))
d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
d_n = torch.zeros(S, dtype=torch.int32, device=dev)
ch = g.dmr_chain(decim, proto, c["center_freq"], c["fs"], gain, omega, c4["gain_omega"], c4["mu"], c4["gain_mu"],
                 c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
ch.set_mode(mode)
st = torch.cuda.Stream(device=dev)
out = {}
for cpw in (1, 8, 32):
    ch.set_captures_per_wave(cpw)
    d_bits.zero_(); d_n.zero_()
    torch.cuda.synchronize()
    ch.run_device(d_in, n, stride, d_bits, n_out, d_n, st)
    st.synchronize()
    nb = d_n.cpu().numpy().copy()
    p_soft, s_soft = ch.intermediate(1)
    softs = []
    for s in range(S):
        soft = np.empty(nb[s], np.float32)
        g.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft), int(nb[s]) * 4)
        softs.append(soft)
    out[cpw] = (nb, softs)
for cpw in (8, 32):
    for s in range(S):
        a1, a2 = out[1][1][s], out[cpw][1][s]
        m = min(len(a1), len(a2))
        d = np.nonzero(a1[:m].view(np.uint32) != a2[:m].view(np.uint32))[0]
        if len(a1) != len(a2) or len(d):
            f = int(d[0]) if len(d) else -1
            print("cpw %d capture %d: counts %d / %d, %d symbols differ, first at %d: %s vs %s" % (
                cpw, s, len(a1), len(a2), len(d), f, a1[max(0, f - 2):f + 4], a2[max(0, f - 2):f + 4]))
    print("cpw %d compared" % cpw)
