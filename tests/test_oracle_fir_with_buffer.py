"""Oracle restatement of gri_fir_filter_with_buffer_{ccf,ccc,fff} (oracle/grdmr_oracle.c orc_fwb_*) pinned the way
the reference's own QA pins the block (filter/qa_gri_fir_filter_with_buffer_ccf.cc:103-170, t1/t2/t3 =
decimate 1, 2, 5): ntaps in [0, 9], lengths in [0, 17], integer-valued data, an explicit delay line as the
expected value, tolerance |expected| * 1e-5 (the reference's ERR_DELTA).  The reference draws its data from glibc
random() after srandom(0); a numpy generator of the same shape is used here (parity of the PROCEDURE, the vectors
themselves are not stored anywhere in the reference)."""
import numpy as np
import pytest


def _data(rng, kind, n, complex_):
    if complex_:
        return (np.rint(rng.uniform(-1, 1, n) * 32767) + 1j * np.rint(rng.uniform(-1, 1, n) * 32767)).astype(np.complex64)
    return np.rint(rng.uniform(-1, 1, n) * 32767).astype(np.float32)


@pytest.mark.parametrize("kind", ["ccf", "ccc", "fff"])
@pytest.mark.parametrize("decimate", [1, 2, 5])
def test_delay_line_procedure(po, kind, decimate):
    rng = np.random.default_rng(decimate * 10 + len(kind))
    MAX_TAPS, OUTPUT_LEN = 9, 17
    INPUT_LEN = MAX_TAPS + OUTPUT_LEN
    for n in range(MAX_TAPS + 1):
        for ol in range(OUTPUT_LEN + 1):
            x = _data(rng, kind, INPUT_LEN, kind != "fff")
            taps = _data(rng, kind, MAX_TAPS, kind == "ccc")[:n]
            nout = ol // decimate
            dline = np.zeros(INPUT_LEN, dtype=np.complex128)
            expected = []
            for o in range(nout):
                for dd in range(decimate):
                    dline[1:] = dline[:-1].copy()
                    dline[0] = x[decimate * o + dd]
                expected.append(np.dot(dline[:n], taps.astype(np.complex128)))      # ref_dotprod: sum input[i] * taps[i]
            f = po.FirFilterWithBuffer(kind, taps)
            got = f.filterNdec(x, nout, decimate)
            for o in range(nout):
                assert abs(complex(got[o]) - expected[o]) <= abs(expected[o]) * 1e-5, (n, ol, o)


@pytest.mark.parametrize("kind", ["ccf", "ccc", "fff"])
def test_equals_the_history_fir_and_continues_across_calls(po, kind):
    """same numbers as gr_fir_XXX on the stream with ntaps-1 zeros in front (up to summation order), any chunking"""
    rng = np.random.default_rng(3)
    T, D, n = 37, 3, 400
    x = _data(rng, kind, n * D, kind != "fff") / np.float32(32767)
    taps = (_data(rng, kind, T, kind == "ccc") / np.float32(32767)).astype(np.complex64 if kind == "ccc" else np.float32)
    whole = po.FirFilterWithBuffer(kind, taps).filterNdec(x, n, D)
    f = po.FirFilterWithBuffer(kind, taps)
    parts = [f.filterNdec(x[a * D:], b - a, D) for a, b in ((0, 1), (1, 50), (50, 51), (51, 399), (399, 400))]
    assert np.array_equal(np.concatenate(parts).view(np.uint8), whole.view(np.uint8))
    hist = np.concatenate([np.zeros(T - 1, x.dtype), x])
    ref = getattr(po, "fir_" + kind)(taps, hist[D - 1:], n, D)
    assert np.abs(whole - ref).max() <= 1e-5 * np.abs(ref).max()
    f.set_taps(taps[:5])                         # set_taps clears the delay line
    again = f.filterNdec(x, 10, 1)
    assert np.array_equal(again.view(np.uint8), po.FirFilterWithBuffer(kind, taps[:5]).filterNdec(x, 10, 1).view(np.uint8))
