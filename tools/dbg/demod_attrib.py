#!/usr/bin/env python3
"""Attribution of the FAST demodulator's per-element deviation on cfg2.

Every figure is measured against ONE yardstick: the reference's own formula evaluated in float64 --
the reference's binary32 composite taps (float(i * fwT0) angle quantisation included) and rotator, the FIR
sum and the demodulator's complex product in float64, gr_fast_atan2f's table (binary32 entries) interpolated in
float64.  Against that yardstick stand: the reference's generic-order build (oracle), the reference's SSE
build (oracle/_ref, where present), and each GPU engine.  "A vs B" rows are the direct comparisons the tests
assert.  Figures: max and rms of |a - b| / |b| over steady-state samples with |b| > 0.1 max|b|, samples on
the reference's arctangent step (z = 1/255) removed as in parity_util.demod_report."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import grhip_loader  # noqa: E402
from parity_util import ATAN_STEP  # noqa: E402


def atan_table():
    txt = open(os.path.join(ROOT, "oracle", "atan_table.inc")).read()
    import re
    words = re.findall(r"0x[0-9a-fA-F]{8}", txt)
    return np.array([int(w, 16) for w in words], dtype=np.uint32).view(np.float32).astype(np.float64)


def fast_atan2_f64(y, x, tab):
    """gr_fast_atan2f (general/gr_fast_atan2f.cc:125-198) with every operation in float64"""
    ya, xa = np.abs(y), np.abs(x)
    big = xa > ya
    num = np.where(big, ya, xa)
    den = np.where(big, xa, ya)
    z = num / np.where(den == 0, 1.0, den)
    alpha = z * 256.0 - 0.5
    idx = np.clip(alpha.astype(np.int64), 0, 255)
    alpha = alpha - idx
    interp = tab[idx] + (tab[idx + 1] - tab[idx]) * alpha
    base = np.where(z < 0.003921569, z, interp)
    q = np.where(big, np.where(x >= 0, 0.0, np.pi), np.pi / 2)
    sb = np.where(big != (x >= 0), -base, base)
    ang = q + sb
    ang = np.where(y >= 0, ang, -ang)
    return np.where(den == 0, 0.0, ang)


def rel_figs(a, b, gain, skip=64):
    a = np.asarray(a, np.float64)[skip:]
    b = np.asarray(b, np.float64)[skip:]
    err = np.abs(a - b)
    step = abs(gain) * ATAN_STEP
    at_step = (err > 1e-5 * np.abs(b).max()) & (np.abs(err - step) <= 0.02 * step)
    big = (~at_step) & (np.abs(b) > 0.1 * np.abs(b).max())
    r = err[big] / np.abs(b[big])
    return float(r.max()), float(np.sqrt((r * r).mean())), int(at_step.sum())


def main():
    n = int(os.environ.get("N", "2000000"))
    g = grhip_loader.import_grhip()
    po = grhip_loader.import_oracle()
    wl = g.workload
    c = wl.CFG2
    D = c["decim"]
    x = wl.fsk4_capture(n, stream_id=11)
    proto = wl.cfg2_proto_taps()
    nout = n // D
    xl = po.Xlating(D, proto, c["center_freq"], c["fs"])
    ct = xl.ctaps().astype(np.complex128)              # the reference's binary32 composite taps
    rot = po.rotator_phases(xl.rot()[1], nout).astype(np.complex128) if hasattr(po, "rotator_phases") else None
    xh = wl.with_history(x, len(proto) - 1).astype(np.complex128)
    # y64[m] = sum_i ct[i] xh[m D + i]: correlation at stride D
    from scipy.signal import fftconvolve
    full = fftconvolve(xh, ct[::-1], mode="valid")      # full[k] = sum_i ct[i] xh[k + i]
    y64 = full[::D][:nout]
    if rot is not None:
        y64 = y64 * rot[:nout]
    prev = np.concatenate([[0.0 + 0.0j], y64[:-1]])
    p = y64 * np.conj(prev)
    tab = atan_table()
    d64 = c["demod_gain"] * fast_atan2_f64(p.imag, p.real, tab)
    gain = c["demod_gain"]

    rows = {}
    rows["reference generic order (oracle)"] = po.chain_xlating_demod(D, proto, c["center_freq"], c["fs"], gain, x)
    if po.have_ref():
        rows["reference SSE build (oracle/_ref)"] = po.chain_xlating_demod(D, proto, c["center_freq"], c["fs"], gain, x, lib="ref")
    xin = wl.with_history(x, len(proto) - 1)
    for name, mode in (("GPU FAST (matrix cores)", g.MODE_FAST), ("GPU FAST_REFTAPS (+ tap-angle quantisation)", g.MODE_FAST_REFTAPS),
                       ("GPU FAST_VALU (f32 vector)", g.MODE_FAST_VALU), ("GPU GENERIC (bit-exact order)", g.MODE_GENERIC)):
        blk = g.xlating_demod(D, proto, c["center_freq"], c["fs"], gain)
        blk.set_mode(mode)
        rows[name] = blk.work(nout, xin)
    if os.environ.get("COMPLEX_TAPS"):
        # probe (iii): the complex-tap (non-pre-mix) engines -- a prototype whose imaginary parts are 1e-30 is "complex" for the
        # dispatch and the same filter for the arithmetic; measured against the same yardstick and reference outputs
        proto_c = (proto.real + 1j * np.float32(1e-30)).astype(np.complex64)
        for name, mode in (("GPU FAST, complex-tap path", g.MODE_FAST), ("GPU FAST_VALU, complex-tap path", g.MODE_FAST_VALU)):
            blk = g.xlating_demod(D, proto_c, c["center_freq"], c["fs"], gain)
            blk.set_mode(mode)
            rows[name] = blk.work(nout, xin)
    print("cfg2, %d samples; per element over |ref| > 0.1 max: max / rms / samples on the arctangent step" % n)
    for k, v in rows.items():
        print("  %-44s vs float64 yardstick: %.3e / %.3e / %d" % ((k,) + rel_figs(v, d64, gain)))
    ref_g = rows["reference generic order (oracle)"]
    for k, v in rows.items():
        if v is not ref_g:
            print("  %-44s vs reference generic:  %.3e / %.3e / %d" % ((k,) + rel_figs(v, ref_g, gain)))
    if "reference SSE build (oracle/_ref)" in rows:
        ref_s = rows["reference SSE build (oracle/_ref)"]
        for k, v in rows.items():
            if k.startswith("GPU"):
                print("  %-44s vs reference SSE:      %.3e / %.3e / %d" % ((k,) + rel_figs(v, ref_s, gain)))


if __name__ == "__main__":
    main()
