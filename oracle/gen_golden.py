#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the pieces of the reference that compile in
the build container (oracle/_ref/libgrref.so, built by oracle/Makefile from
/root/reference).  The fixtures are DATA (inputs + reference outputs); they are
committed because the reference cannot travel to the GPU box.

Also writes tests/golden/ref_qa_vectors.json: constants transcribed from the
reference's own unit tests (qa_fft.py, qa_correlate_access_code.py,
qa_clock_recovery_mm.py, qa_gr_fir_fff.cc) -- test data, not code.

Run:  make -C oracle && python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import pyoracle as po  # noqa: E402
import grhip_loader  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def atan_fixture():
    # signed axis values including zeros, -0.0, denormal-ish, equal magnitudes
    base = np.concatenate([
        np.array([0.0, -0.0, 1e-30, 1e-8, 1e-4, 0.003, 0.00392, 0.003922, 0.01, 0.5, 1.0, 2.0, 255.0, 256.0, 1e6],
                 dtype=np.float32),
        np.random.default_rng(1).uniform(0, 4, 49).astype(np.float32)])
    ax = np.concatenate([base, -base]).astype(np.float32)
    yy, xx = np.meshgrid(ax, ax, indexing="ij")
    out = po.ref_fast_atan2f(yy.ravel(), xx.ravel()).reshape(yy.shape)
    return dict(axis=ax, out=out)


def rotator_fixture():
    incs = []
    outs = []
    for ang in (0.7853981852531433 * 4, 0.1, -2.5):
        inc = np.complex64(np.exp(1j * float(np.float32(ang))))
        incs.append(inc)
        outs.append(po.ref_rotator_phases(inc, 4096))
    return dict(incr=np.array(incs, dtype=np.complex64), phases=np.stack(outs))


def clip_fixture():
    x = np.concatenate([np.linspace(-1, 1, 401), [0.5, 0.001, -0.001, 0.0049999, 0.005, 0.0050001]]).astype(np.float32)
    return dict(x=x, clip=np.float32(0.005), out=po.ref_branchless_clip(x, 0.005),
                clip2=np.float32(0.001), out2=po.ref_branchless_clip(x, 0.001))


def bits_fixture():
    rng = np.random.default_rng(2)
    w = rng.integers(0, 2 ** 63, 512, dtype=np.uint64) * 2 + rng.integers(0, 2, 512, dtype=np.uint64)
    w[:4] = [0, 0xFFFFFFFFFFFFFFFF, 0xF0F0F0F0F0F0F0F1, 1]
    return dict(words=w, counts=po.ref_count_bits64(w))


def sse_fir_fixture():
    """reference SSE dot-product kernels behind the alignment logic of
    gr_fir_*_simd, for every (kind, ntaps, decim) and the 2 (complex) / 4
    (float) input alignments; integer-valued data (exact in any summation
    order) and float data (tolerance target)."""
    rng = np.random.default_rng(3)
    d = {}
    n = 40
    for kind in ("fff", "ccf", "ccc"):
        for ntaps in (1, 2, 7, 8, 64, 255, 256):
            for decim in (1, 4):
                nin = (n - 1) * decim + ntaps
                for dat in ("int", "flt"):
                    if dat == "int":
                        mk = lambda m: rng.integers(-7, 8, m).astype(np.float32)
                    else:
                        mk = lambda m: rng.uniform(-1, 1, m).astype(np.float32)
                    if kind == "fff":
                        x = mk(nin); taps = mk(ntaps)
                    elif kind == "ccf":
                        x = (mk(nin) + 1j * mk(nin)).astype(np.complex64); taps = mk(ntaps)
                    else:
                        x = (mk(nin) + 1j * mk(nin)).astype(np.complex64)
                        taps = (mk(ntaps) + 1j * mk(ntaps)).astype(np.complex64)
                    key = "%s_%d_%d_%s" % (kind, ntaps, decim, dat)
                    d[key + "_x"] = x
                    d[key + "_t"] = taps
                    nal = 4 if kind == "fff" else 2
                    d[key + "_y"] = np.stack([po.ref_fir_sse(kind, taps, x, n, decim, misalign=a) for a in range(nal)])
    return d


def chain_fixture():
    g = grhip_loader.import_grhip()
    wl = g.workload
    c = wl.CFG2
    n = 48_000
    x = wl.fsk4_capture(n, stream_id=77)
    proto = wl.cfg2_proto_taps()
    y, dem = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x,
                                    want_y=True, lib="ref")
    return dict(stream_id=np.int64(77), n=np.int64(n), y=y, demod=dem)


def qa_vectors():
    primes = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101,
              103, 107, 109, 113, 127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199,
              211, 223, 227, 229, 233, 239, 241, 251, 257, 263, 269, 271, 277, 281, 283, 293, 307, 311)
    fft32 = [
        (4377, 4516), (-1706.1268310546875, 1638.4256591796875), (-915.2083740234375, 660.69427490234375),
        (-660.370361328125, 381.59600830078125), (-499.96044921875, 238.41630554199219),
        (-462.26748657226562, 152.88948059082031), (-377.98440551757812, 77.5928955078125),
        (-346.85821533203125, 47.152004241943359), (-295, 20), (-286.33609008789062, -22.257017135620117),
        (-271.52999877929688, -33.081821441650391), (-224.6358642578125, -67.019538879394531),
        (-244.24473571777344, -91.524826049804688), (-203.09068298339844, -108.54627227783203),
        (-198.45195007324219, -115.90768432617188), (-182.97744750976562, -128.12318420410156),
        (-167, -180), (-130.33688354492188, -173.83778381347656), (-141.19784545898438, -190.28807067871094),
        (-111.09677124023438, -214.48896789550781), (-70.039543151855469, -242.41630554199219),
        (-68.960540771484375, -228.30015563964844), (-53.049201965332031, -291.47097778320312),
        (-28.695289611816406, -317.64553833007812), (57, -300), (45.301143646240234, -335.69509887695312),
        (91.936195373535156, -373.32437133789062), (172.09465026855469, -439.275146484375),
        (242.24473571777344, -504.47515869140625), (387.81732177734375, -666.6788330078125),
        (689.48553466796875, -918.2142333984375), (1646.539306640625, -1694.1956787109375)]
    return {
        "source": {
            "fft": "gnuradio-core/src/python/gnuradio/gr/qa_fft.py:40-153 (rel_eps 4e-4, abs_eps 1e-9)",
            "corr": "gr-digital/python/qa_correlate_access_code.py:27-78",
            "mm": "gr-digital/python/qa_clock_recovery_mm.py:70-102,140-172",
            "fir_fff": "gnuradio-core/src/lib/filter/qa_gr_fir_fff.cc:58-76",
        },
        "fft32": {"primes": list(primes), "forward_expected": fft32, "rel_eps": 4e-4, "abs_eps": 1e-9},
        "corr": {
            "default_access_code_bytes": [0xAC, 0xDD, 0xA4, 0xE2, 0xF2, 0x8C, 0x20, 0xFC],
            "test_001": {"code": "1011", "threshold": 0,
                         "src": [1, 0, 1, 1, 1, 1, 0, 1, 1] + [0] * 64 + [0] * 7,
                         "expected": [0] * 64 + [1, 0, 1, 1, 3, 1, 0, 1, 1, 2] + [0] * 6},
            "test_002": {"tail": [1, 0, 1, 1], "expected_tail": [3, 0, 1, 1]},
        },
        "mm": {
            "test02": {"omega": 2, "gain_omega": 0.01, "mu": 0.5, "gain_mu": 0.01, "omega_rel_lim": 0.001,
                       "data": "100*[1]", "expected_last30": 0.99972, "places": 5},
            "test04": {"omega": 2, "gain_omega": 0.01, "mu": 0.25, "gain_mu": 0.1, "omega_rel_lim": 0.001,
                       "data": "1000*[1,1,-1,-1]", "expected_last100": [-1.31, 1.31], "places": 1},
        },
        "fir_fff": {"input": [234, -4, 23, -56, 45, 98, -23, -7], "taps_1a": [-3],
                    "expected_1a": [-702, 12, -69, 168, -135, -294, 69, 21], "taps_1b": [-4, 5],
                    "expected_1b": [1186, -112, 339, -460, -167, 582, -87]},
    }


def main():
    if not po.have_ref():
        raise SystemExit("oracle/_ref/libgrref.so missing: run `make -C oracle` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "ref_atan2.npz"), **atan_fixture())
    np.savez_compressed(os.path.join(GOLD, "ref_rotator.npz"), **rotator_fixture())
    np.savez_compressed(os.path.join(GOLD, "ref_clip.npz"), **clip_fixture())
    np.savez_compressed(os.path.join(GOLD, "ref_count_bits.npz"), **bits_fixture())
    np.savez_compressed(os.path.join(GOLD, "ref_mmse_taps.npz"), taps=po.ref_mmse_taps())
    np.savez_compressed(os.path.join(GOLD, "ref_sse_fir.npz"), **sse_fir_fixture())
    np.savez_compressed(os.path.join(GOLD, "ref_chain_cfg2.npz"), **chain_fixture())
    with open(os.path.join(GOLD, "ref_qa_vectors.json"), "w") as f:
        json.dump(qa_vectors(), f, indent=1)
    for fn in sorted(os.listdir(GOLD)):
        print("%8d  %s" % (os.path.getsize(os.path.join(GOLD, fn)), fn))


if __name__ == "__main__":
    main()
