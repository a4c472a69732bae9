"""Secondary measurement (not the headline bench line): the full DMR chain
(config 4/5 shape) over a batch of S captures on one GPU, per-kernel times via
torch events.  usage: python tools/bench_chain.py [S] [n_samples] [decim ntaps]
(decim / ntaps given: a low-pass of that length in front of that decimation, the symbol clock rescaled to the new rate)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import grhip_loader

CPW = None
FOUR = "--four" in sys.argv          # the 4FSK tail (pager slicer -> dibits -> correlator) instead of the binary slicer
if FOUR:
    sys.argv.remove("--four")
GENERIC = "--generic" in sys.argv    # the whole chain in GRHIP_MODE_GENERIC (bit-exact)
if GENERIC:
    sys.argv.remove("--generic")
if "--cpw" in sys.argv:          # captures per wave of the clock recovery: 1 / 8 (default: the library's choice)
    i = sys.argv.index("--cpw")
    CPW = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
g = grhip_loader.import_grhip()
wl = g.workload
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
c, c4 = wl.CFG2, wl.CFG4
decim = int(sys.argv[3]) if len(sys.argv) > 4 else c["decim"]
proto = wl.lowpass_taps(int(sys.argv[4]), 100e3, 10e6).astype(np.complex64) if len(sys.argv) > 4 else wl.cfg2_proto_taps()
omega = c4["omega"] * c["decim"] / decim
dev = torch.device("cuda", 0)
# DISTINCT captures, one stream id each, synthesised on the device like bench.py's (S copies of one capture
# step every clock-recovery loop in lock-step, the friendliest case for eight captures per wavefront)
import importlib.util
_spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
d_in = torch.empty((S, n, 2), dtype=torch.float32, device=dev)
for k0 in range(0, S, 64):
    kk = min(64, S - k0)
    d_in[k0:k0 + kk] = _bench.synth_captures(torch, wl, kk, n, 1000 + k0, dev)
torch.cuda.synchronize()
nout = n // decim
d_bits = torch.zeros((S, nout), dtype=torch.uint8, device=dev)
d_n = torch.zeros(S, dtype=torch.int32, device=dev)
ch = g.dmr_chain(decim, proto, c["center_freq"], c["fs"], c["demod_gain"], omega,
                 c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], wl.access_code_string(),
                 c4["threshold"], S, n)
if GENERIC:
    ch.set_mode(g.MODE_GENERIC)
if CPW is not None:
    ch.set_captures_per_wave(CPW)
if FOUR:
    ch.set_four_level(True, 0.001)
    d_bits = torch.zeros((S, 2 * nout), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev)
for _ in range(2):
    ch.run_device(d_in, n, n, d_bits, (2 if FOUR else 1) * nout, d_n, st)
st.synchronize()
reps = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    ch.run_device(d_in, n, n, d_bits, (2 if FOUR else 1) * nout, d_n, st)
e1.record(st)
st.synchronize()
ms = e0.elapsed_time(e1) / reps
print(json.dumps({"workload": "full DMR chain (xlating+demod -> M&M -> slicer+correlator)", "streams": S, "four_level": FOUR, "mode": "GENERIC" if GENERIC else "FAST", "captures_per_wave": CPW, "decim": decim,
                  "ntaps": len(proto), "samples_per_stream": n, "ms_per_batch": ms, "Msamples_per_s": S * n / ms / 1e3,
                  "captures": "distinct stream ids 1000 ... %d" % (999 + S),
                  "symbols_min_max": [int(d_n.min().item()), int(d_n.max().item())],
                  "access_code_flags": sum(int(torch.count_nonzero(d_bits[r0:r0 + 64] & 2).item()) for r0 in range(0, d_bits.shape[0], 64))}))
del ch        # (streams and events released before the interpreter tears the runtime down)
torch.cuda.synchronize()
