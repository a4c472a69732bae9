"""digital_clock_recovery_mm_cc through the C ABI: bit-exact against the oracle, whose
restatement is pinned by the reference's own QA (tests/test_oracle_mmcc.py)."""
import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu


def qpsk(rng, nsym, sps, noise=0.1):
    sym = (rng.integers(0, 2, nsym) * 2 - 1) + 1j * (rng.integers(0, 2, nsym) * 2 - 1)
    x = np.repeat(sym, sps).astype(np.complex64)
    k = np.ones(sps) / sps
    x = np.convolve(x, k)[:len(x)]
    x = x + noise * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))
    return x.astype(np.complex64)


def drive(blk, buf, nout, want_error):
    outs, errs, pos = [], [], 0
    while True:
        y, e, c = blk.general_work(nout, buf[pos:], want_error)
        if len(y) == 0:
            break
        outs.append(np.array(y))
        if want_error:
            errs.append(np.array(e))
        pos += c
    cat = lambda a, dt: np.concatenate(a) if a else np.zeros(0, dt)
    return cat(outs, np.complex64), cat(errs, np.float32), pos


@pytest.mark.parametrize("want_error", [False, True])
@pytest.mark.parametrize("params,sps,nout", [((2.0, 0.001, 0.5, 0.01, 0.001), 2, 300), ((4.0, 0.25 * 0.175 ** 2, 0.5, 0.175, 0.005), 4, 4096),
                                              ((8.3, 0.01, 0.1, 0.1, 0.01), 8, 1000), ((2.0, 0.01, 0.25, 0.1, 0.0001), 2, 77)])
def test_mmcc_bit_exact_against_oracle(gpu, po, params, sps, nout, want_error):
    rng = np.random.default_rng(int(params[0] * 10) + nout)
    x = np.concatenate([np.zeros(2, np.complex64), qpsk(rng, 6000, sps)])         # history 3
    blk, ref = gpu.clock_recovery_mm_cc(*params), po.ClockRecoveryMMcc(*params)
    assert blk.history() == 3 and blk.forecast(nout) == ref.forecast(nout)
    y, e, used = drive(blk, x, nout, want_error)
    yr, er, usedr = drive(ref, x, nout, want_error)
    assert used == usedr and len(y) == len(yr) and len(y) > 100
    assert bits_equal(y, yr)
    if want_error:
        assert bits_equal(e, er)
    assert blk.mu().tobytes() == ref.mu().tobytes() and blk.omega().tobytes() == ref.omega().tobytes()
    assert blk.forecast(nout) == ref.forecast(nout)                                # omega has moved


def test_mmcc_reference_qa_vectors(gpu, po):
    """qa_clock_recovery_mm.py:35-67 (test01) and 104-137 (test03) on the GPU path"""
    blk = gpu.clock_recovery_mm_cc(2, 0.001, 0.5, 0.01, 0.001)
    y, _, _ = drive(blk, np.concatenate([np.zeros(2), 100 * [1 + 1j]]).astype(np.complex64), 512, False)
    assert np.allclose(y[-30:], 0.99972 + 0.99972j, atol=5e-6)
    blk = gpu.clock_recovery_mm_cc(2, 0.01, 0.25, 0.1, 0.0001)
    data = np.array([0, 0] + 1000 * [1 + 1j, 1 + 1j, -1 - 1j, -1 - 1j], dtype=np.complex64)
    y, _, _ = drive(blk, data, 512, False)
    t = y[-100:]
    assert np.abs(np.abs(t.real) - 1.2).max() < 0.05 and np.abs(np.abs(t.imag) - 1.2).max() < 0.05
    assert np.all(np.sign(t.real[1:]) == -np.sign(t.real[:-1]))                   # alternating, as expected_result


def test_mmcc_setters_and_errors(gpu, po):
    rng = np.random.default_rng(3)
    x = np.concatenate([np.zeros(2, np.complex64), qpsk(rng, 3000, 4)])
    p = (4.0, 0.01, 0.5, 0.1, 0.01)
    blk, ref = gpu.clock_recovery_mm_cc(*p), po.ClockRecoveryMMcc(*p)
    y1, _, c1 = blk.general_work(500, x)
    r1, _, d1 = ref.general_work(500, x)
    assert c1 == d1 and bits_equal(np.array(y1), r1)
    blk.set_mu(0.25); blk.set_gain_mu(0.05)
    assert blk.mu() == np.float32(0.25) and blk.gain_mu() == np.float32(0.05)
    for bad in [(0.0, 0.1, 0.5, 0.1, 0.01), (2.0, -0.1, 0.5, 0.1, 0.01), (2.0, 0.1, 0.5, -0.1, 0.01)]:
        with pytest.raises(gpu.GrhipError):
            gpu.clock_recovery_mm_cc(*bad)
        with pytest.raises(IndexError):
            po.ClockRecoveryMMcc(*bad)
    # not enough input: produces nothing, consumes nothing (.cc:130: ni <= 0)
    y, _, c = gpu.clock_recovery_mm_cc(*p).general_work(100, x[:20])
    assert len(y) == 0 and c == 0
