"""CPU check of the matrix-core FIR engine's host geometry (csrc/mfma_tables.h): the banded
Toeplitz operand, the split into two binary16 halves, the tile/plane arithmetic.  The header
is plain C++; a small g++ program evaluates it and numpy checks the results."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "gnuradio-3.5.0-dmr_amd", "csrc")

PROG = r"""
#include <cstdio>
#include <cstdlib>
#include "mfma_tables.h"
using namespace grhip;
int main(int argc, char **argv)
{
    int D = atoi(argv[1]), T = atoi(argv[2]), off = atoi(argv[3]);
    std::vector<float> h(T);
    unsigned s = 12345u + T * 31 + D;
    for (int i = 0; i < T; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 20001 - 10000) * 1e-6f * (1 + (i % 7)); }
    int KS = mf::ksteps_inst(D, T), kexp = mf::tap_scale_exp(h.data(), T);
    std::vector<uint16_t> A;
    mf::build_A(h.data(), T, D, KS, off, kexp, A);
    printf("%d %d %d %d %d %d %d %d\n", KS, kexp, mf::tile_samples(D, KS), mf::rounds(D, KS), mf::plane_bytes(D, KS),
           mf::chunks_per_seg(D, KS), (int)mf::supported(D, T), mf::ksteps_for(D, T));
    for (int i = 0; i < T; ++i) printf("%.9g\n", h[i]);
    for (size_t i = 0; i < A.size(); ++i) printf("%.9g\n", mf::f16_to_f32(A[i]));
    // conversion spot checks
    float probes[] = {0.f, 1.f, -1.f, 65504.f, 65519.f, 65520.f, 1e-8f, 5.96046448e-8f, 6.1035e-5f, 0.333333f, 2049.f, 2051.f};
    for (float p : probes) printf("%.9g\n", mf::f16_to_f32(mf::f32_to_f16(p)));
    return 0;
}
"""


@pytest.fixture(scope="module")
def prog(tmp_path_factory):
    d = tmp_path_factory.mktemp("mfma")
    src = d / "t.cc"
    src.write_text(PROG)
    exe = d / "t"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", HDR, str(src), "-o", str(exe)])
    return str(exe)


def _run(prog, D, T, off):
    out = subprocess.check_output([prog, str(D), str(T), str(off)], text=True).split("\n")
    head = [int(v) for v in out[0].split()]
    vals = np.array([float(v) for v in out[1:] if v.strip()])
    return head, vals


@pytest.mark.parametrize("D,T,off", [(4, 256, 0), (4, 256, 1), (4, 100, 1), (2, 256, 0), (2, 130, 1), (4, 259, 1), (2, 289, 0)])
def test_band_product_is_the_fir(prog, D, T, off):
    (KS, kexp, SP, NI, PL, NCH, sup, ks_need), vals = _run(prog, D, T, off)
    assert sup == 1 and ks_need <= KS
    h = vals[:T]
    A = vals[T:T + KS * 2 * 64 * 8].reshape(KS, 2, 64, 8)
    # lane order of v_mfma_f32_16x16x32_f16: lane l holds A[row l & 15][k = 8 (l >> 4) + j]
    M = np.zeros((2, 16, KS * 32))
    for js in range(KS):
        for l in range(64):
            M[:, l & 15, 32 * js + 8 * (l >> 4): 32 * js + 8 * (l >> 4) + 8] = A[js, :, l, :]
    full = (M[0] + M[1]) * 2.0 ** (-kexp)
    # the band: row a holds h at k = D a + off + i
    want = np.zeros((16, KS * 32))
    for a in range(16):
        want[a, D * a + off: D * a + off + T] = h
    assert np.abs(full - want).max() <= 2.0 ** -21 * np.abs(h).max()        # two halves = ~22 bits
    assert np.abs(M[0]).max() < 2 ** 14 and np.abs(M[0]).max() >= 2 ** 13    # the scaling rule
    # the product against a random stream segment is the decimating FIR
    rng = np.random.default_rng(T + D)
    x = rng.standard_normal(KS * 32 + 8)
    y = full @ x[:KS * 32]
    ref = np.array([np.dot(h, x[D * a + off: D * a + off + T]) for a in range(16)])
    assert np.abs(y - ref).max() <= 1e-6 * np.abs(h).sum() * np.abs(x).max()


def test_geometry(prog):
    (KS, kexp, SP, NI, PL, NCH, sup, _), vals = _run(prog, 4, 256, 0)
    assert (KS, SP, NI, NCH) == (10, 8256, 17, 16)
    assert PL % 256 == 0 and PL >= 2 * SP + 32 * (SP // 256)
    assert 4 * PL + 4 * 8 * 36 * 4 + 2048 + 32 <= 80 * 1024              # two workgroups per CU
    (KS, kexp, SP, NI, PL, NCH, sup, _), _v = _run(prog, 2, 256, 0)
    assert (KS, NCH) == (10, 13) and SP == (3 * 496 + 7 * 64) * 2 + 13 * 32
    # shapes the engine does not take
    assert _run(prog, 4, 261, 0)[0][6] == 0 and _run(prog, 2, 291, 0)[0][6] == 0
    probes = vals[-12:]
    assert list(probes[:7]) == [0.0, 1.0, -1.0, 65504.0, 65504.0, np.inf, 0.0]
    assert abs(probes[7] - 2.0 ** -24) < 1e-15 and abs(probes[8] - 6.1035e-5) < 3e-8
    assert probes[10] == 2048.0 and probes[11] == 2052.0                      # ties to even
