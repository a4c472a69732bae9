"""gr_pfb_decimator_ccf through the C ABI against the oracle's restatement of
filter/gr_pfb_decimator_ccf.cc:77-180 and against the formula evaluated in double.  The reference has no QA
for this block and gets the channel sum from FFTW: tolerance 1e-5 of the output's peak (BASELINE north_star),
rounding unpinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-5


def streams_for(rng, M, tpf, n):
    return [(rng.standard_normal(n + tpf - 1) + 1j * rng.standard_normal(n + tpf - 1)).astype(np.complex64)
            for _ in range(M)]


def direct_double(M, taps, chan, streams, n):
    tpf = -(-len(taps) // M)
    tp = np.zeros(M * tpf)
    tp[:len(taps)] = taps
    out = np.zeros(n, dtype=np.complex128)
    for j in range(M):
        h = tp[j::M]                                         # d_taps[j][t] = taps[j + t*M]
        x = streams[M - 1 - j].astype(np.complex128)
        f = np.convolve(x, h)[tpf - 1:tpf - 1 + n]           # sum_t h[t] x[i + tpf-1 - t]
        out += f * np.exp(2j * np.pi * j * chan / M)
    return out


@pytest.mark.parametrize("M,ntaps,chan,n", [(8, 256, 3, 5000), (1, 16, 0, 1500), (2, 7, 1, 1024), (3, 100, 2, 1023),
                                              (16, 200, 5, 1025), (20, 400, 19, 3000), (5, 3, 4, 100), (8, 2048, 0, 2500)])
def test_pfb_decimator_matches_oracle_and_formula(gpu, po, wl, M, ntaps, chan, n):
    rng = np.random.default_rng(M * 1000 + ntaps)
    taps = wl.lowpass_taps(ntaps, 0.4 / M, 1.0) if ntaps > 8 else rng.standard_normal(ntaps).astype(np.float32)
    blk = gpu.pfb_decimator_ccf(M, taps, chan)
    tpf = -(-ntaps // M)
    assert blk.history() == tpf
    xs = streams_for(rng, M, tpf, n)
    assert len(blk.work(n, xs)) == 0                         # d_updated: returns 0 once (.cc:138-141)
    y = blk.work(n, xs)
    assert len(y) == n
    ref = po.pfb_decimator_ccf(M, taps, chan, xs, n)
    dd = direct_double(M, taps, chan, xs, n)
    scale = np.abs(dd).max()
    assert np.abs(ref - dd).max() <= TOL * scale             # the oracle against the formula
    assert np.abs(y - ref).max() <= TOL * scale
    assert np.abs(y - dd).max() <= TOL * scale


def test_pfb_decimator_set_taps_and_device_path(gpu, po, wl):
    import torch
    rng = np.random.default_rng(5)
    M, n = 8, 300_000
    t1, t2 = wl.lowpass_taps(128, 0.05, 1.0), wl.lowpass_taps(77, 0.04, 2.0)
    blk = gpu.pfb_decimator_ccf(M, t1, 6)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    for taps in (t1, t2):
        if taps is t2:
            blk.set_taps(taps)
        tpf = blk.history()
        xs = streams_for(rng, M, tpf, n)
        per = n + tpf - 1 + 5                                # stride longer than a stream
        flat = np.zeros((M, per), dtype=np.complex64)
        for s in range(M):
            flat[s, :n + tpf - 1] = xs[s]
            flat[s, n + tpf - 1:] = np.nan                   # never read
        d_in = torch.from_numpy(flat.view(np.float32).reshape(M, per, 2)).to(dev)
        d_out = torch.empty((n, 2), device=dev)
        assert blk.work_device(n, d_in, per, d_out, st) == 0
        assert blk.work_device(n, d_in, per, d_out, st) == n
        st.synchronize()
        y = d_out.cpu().numpy().view(np.complex64).reshape(-1)
        ref = po.pfb_decimator_ccf(M, taps, 6, xs, n)
        assert np.abs(y - ref).max() <= TOL * np.abs(ref).max()


def test_pfb_decimator_after_stream_to_streams(gpu, po, wl):
    """blks2impl/pfb_decimator.py: stream_to_streams(decim) -> pfb_decimator_ccf: a decimating band-pass filter"""
    rng = np.random.default_rng(6)
    M, chan, ntaps, n = 4, 1, 64, 20000
    taps = wl.lowpass_taps(ntaps, 0.1, 1.0)
    tpf = ntaps // M
    x = (rng.standard_normal((n + tpf - 1) * M) + 1j * rng.standard_normal((n + tpf - 1) * M)).astype(np.complex64)
    s2s = gpu.stream_to_streams(8, M)
    xs = s2s.work(n + tpf - 1, x)
    blk = gpu.pfb_decimator_ccf(M, taps, chan)
    blk.work(n, xs)
    y = blk.work(n, xs)
    assert np.abs(y - direct_double(M, taps, chan, xs, n)).max() <= TOL * np.abs(y).max()


def test_pfb_decimator_errors(gpu):
    with pytest.raises(gpu.GrhipError):
        gpu.pfb_decimator_ccf(0, [1.0], 0)
