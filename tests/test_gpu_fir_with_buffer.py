"""-m gpu: gri_fir_filter_with_buffer_{ccf,ccc,fff} through the C ABI.  GENERIC mode accumulates in the reference's
order (one accumulator, filter/gri_fir_filter_with_buffer_XXX.cc.t:75-79): bit-exact against the oracle, across
calls and changes of the decimation; FAST mode: 1e-5."""
import numpy as np
import pytest

from conftest import bits_equal, rel_err_max

pytestmark = pytest.mark.gpu


def _data(rng, kind, n, complex_):
    if complex_:
        return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    return rng.uniform(-1, 1, n).astype(np.float32)


@pytest.mark.parametrize("kind", ["ccf", "ccc", "fff"])
@pytest.mark.parametrize("ntaps", [0, 1, 2, 9, 64, 256, 300])
def test_generic_bit_exact_across_calls(gpu, po, kind, ntaps):
    rng = np.random.default_rng(ntaps + len(kind))
    x = _data(rng, kind, 60_000, kind != "fff")
    taps = _data(rng, kind, ntaps, kind == "ccc")
    blk = gpu.fir_filter_with_buffer(kind, taps)
    blk.set_mode(gpu.MODE_GENERIC)
    ref = po.FirFilterWithBuffer(kind, taps)
    assert blk.ntaps() == ntaps
    pos = 0
    for n, dec in ((1, 1), (17, 1), (1000, 1), (333, 4), (1, 7), (5000, 2), (2048, 4), (3, 1)):
        seg = x[pos:pos + n * dec]
        assert bits_equal(blk.filterNdec(seg, n, dec), ref.filterNdec(seg, n, dec)), (n, dec)
        pos += n * dec
    blk.set_taps(taps[: ntaps // 2])             # new taps: delay line cleared, at once
    ref.set_taps(taps[: ntaps // 2])
    assert bits_equal(blk.filterN(x[:500], 500), ref.filterNdec(x[:500], 500, 1))


@pytest.mark.parametrize("kind,ntaps,dec", [("ccf", 256, 4), ("ccf", 64, 1), ("ccc", 100, 2), ("fff", 256, 1), ("fff", 31, 2),
                                            ("ccc", 700, 1)])
def test_fast_mode_tolerance(gpu, po, kind, ntaps, dec):
    rng = np.random.default_rng(ntaps * 3 + dec)
    n = 20_000
    x = _data(rng, kind, 2 * n * dec, kind != "fff")
    taps = _data(rng, kind, ntaps, kind == "ccc")
    blk = gpu.fir_filter_with_buffer(kind, taps)
    ref = po.FirFilterWithBuffer(kind, taps)
    for part in (x[: n * dec], x[n * dec:]):     # second call continues from the first one's delay line
        assert rel_err_max(blk.filterNdec(part, n, dec), ref.filterNdec(part, n, dec)) <= 1e-5


def test_device_entry(gpu, po):
    import torch
    rng = np.random.default_rng(1)
    taps = _data(rng, "ccf", 128, False)
    x = _data(rng, "ccf", 40_000, True)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
    d_y = torch.zeros((10_000, 2), dtype=torch.float32, device=dev)
    blk = gpu.fir_filter_with_buffer("ccf", taps)
    ref = po.FirFilterWithBuffer("ccf", taps)
    for k in range(2):
        blk.filterNdec_device(d_y[k * 5000:], d_x[k * 20_000:], 5000, 4, st)
    st.synchronize()
    got = d_y.cpu().numpy().reshape(-1).view(np.complex64)
    assert rel_err_max(got, ref.filterNdec(x, 10_000, 4)) <= 1e-5
