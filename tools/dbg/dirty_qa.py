"""FIR QA cases (qa_gr_fir_ccf.cc shape: integer-valued data, 0..9 taps, 0..17 outputs) through the kernel-level entries
after the device memory has been filled with garbage and released: looks for reads of memory that was never written"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import grhip_loader
g = grhip_loader.import_grhip()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
def dirty():
    bufs = [torch.full((1 << 28,), 3.0e4, dtype=torch.float32, device=dev) for _ in range(40)]     # 40 GB of 30000.0
    torch.cuda.synchronize()
    del bufs
    torch.cuda.empty_cache()
fails = 0
cases = 0
for rep in range(6):
    dirty()
    for kind in ("ccf", "ccc", "fff"):
        for n in range(0, 10):
            for ol in range(0, 18):
                L = 9 + 17
                if kind == "fff":
                    x = np.rint(rng.uniform(-1, 1, L) * 32768).astype(np.float32)
                    taps = np.rint(rng.uniform(-1, 1, 9) * 32768).astype(np.float32)
                else:
                    x = (np.rint(rng.uniform(-1, 1, L) * 32767) + 1j * np.rint(rng.uniform(-1, 1, L) * 32767)).astype(np.complex64)
                    taps = np.rint(rng.uniform(-1, 1, 9) * 32767).astype(np.float32) if kind == "ccf" else \
                        (np.rint(rng.uniform(-1, 1, 9) * 32767) + 1j * np.rint(rng.uniform(-1, 1, 9) * 32767)).astype(np.complex64)
                t = taps[:n]
                blk = getattr(g, "fir_filter_" + kind)(1, t)
                got = blk.filterNdec(x, ol, 1) if ol else np.zeros(0, x.dtype)
                for o in range(ol):
                    s = sum(complex(x[o + i]) * complex(t[n - i - 1]) for i in range(n))
                    cases += 1
                    if abs(complex(got[o]) - s) > abs(s) * 1e-5:
                        fails += 1
                        print("rep", rep, kind, "ntaps", n, "outputs", ol, "output", o, "got", got[o], "expected", s, flush=True)
                del blk
print("cases", cases, "fails", fails)
