"""Where does the overlap-save engine overtake the tiled vector kernel?  Diagnostic builds only (GRHIP_OLS_MIN):
   make variant NAME=diag EXTRA=-DGRHIP_DIAG; GRHIP_LIB=.../libgrhip_diag.so python tools/bench_ols_crossover.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import grhip_loader
    g = grhip_loader.import_grhip()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(0)
    n_in = 1 << 24
    x = torch.randn((n_in + 4096, 2), device=dev)
    for kind in ("ccf", "ccc"):
        for decim in (1, 2, 4):
            for tpp in (16, 24, 32, 48, 64, 96, 128):
                ntaps = tpp * decim
                if ntaps < 48:
                    continue
                taps = rng.uniform(-1, 1, ntaps).astype(np.float32) if kind == "ccf" else \
                    (rng.uniform(-1, 1, ntaps) + 1j * rng.uniform(-1, 1, ntaps)).astype(np.complex64)
                blk = g.fir_filter_ccf(decim, taps) if kind == "ccf" else g.fir_filter_ccc(decim, taps)
                blk.set_mode(g.MODE_FAST_VALU)          # (the matrix-core engine is not part of this comparison)
                n = n_in // decim
                y = torch.empty((n, 2), device=dev)
                for _ in range(30):
                    blk.work_device(n, x, y, st)
                st.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(20):
                    blk.work_device(n, x, y, st)
                e1.record(st)
                st.synchronize()
                ms = e0.elapsed_time(e1) / 20
                print(json.dumps({"kind": kind, "decim": decim, "taps_per_phase": tpp, "Gsamples_s": round(n_in / ms / 1e6, 1)}), flush=True)
    sys.exit(0)

res = {}
for name, thr in (("tiled", "9999"), ("engine", "0")):
    env = dict(os.environ, GRHIP_OLS_MIN=thr)
    out = subprocess.run([sys.executable, __file__, "child"], env=env, stdout=subprocess.PIPE, text=True).stdout
    for l in out.splitlines():
        if l.startswith("{"):
            d = json.loads(l)
            res.setdefault((d["kind"], d["decim"], d["taps_per_phase"]), {})[name] = d["Gsamples_s"]
for k in sorted(res):
    print("%s D=%d taps/phase %3d: tiled %7.1f  engine %7.1f Gsamples/s" % (k[0], k[1], k[2], res[k].get("tiled", 0), res[k].get("engine", 0)))
