"""-m gpu parity tests: gr_fft_vcc and gr_pfb_channelizer_ccf.
FFT tolerance: the reference's own (qa_fft.py: rel 4e-4) on its 32-point
vectors; against the float64 DFT oracle 1e-6*log2(N) of ||X||inf (FFTW3f is a
third-party float transform whose codelet order is not restatable: "parity
unpinned" beyond the reference's 32-point vectors)."""
import json
import os

import numpy as np
import pytest

from conftest import rel_err_max

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rc(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def test_fft_reference_qa_vectors(gpu):
    with open(os.path.join(GOLD, "ref_qa_vectors.json")) as f:
        v = json.load(f)["fft32"]
    p = v["primes"]
    src = np.array([complex(p[2 * i], p[2 * i + 1]) for i in range(32)], np.complex64)
    exp = np.array([complex(a, b) for a, b in v["forward_expected"]], np.complex64)
    got = gpu.fft_vcc(32, True, [], False).work(1, src)
    assert np.all(np.abs(got - exp) <= v["abs_eps"] + v["rel_eps"] * np.abs(exp))
    back = gpu.fft_vcc(32, False, [], False).work(1, (exp / 32).astype(np.complex64))
    assert np.all(np.abs(back - src) <= v["abs_eps"] + v["rel_eps"] * np.abs(src))


@pytest.mark.parametrize("N", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("forward", [True, False])
def test_fft_sizes_vs_oracle(gpu, po, N, forward):
    rng = np.random.default_rng(N + int(forward))
    nvec = 5
    x = _rc(rng, N * nvec)
    ref = po.fft_vcc(N, forward, None, False, x)
    got = gpu.fft_vcc(N, forward, [], False).work(nvec, x)
    assert rel_err_max(got, ref) <= 1e-6 * max(np.log2(N), 1)


@pytest.mark.parametrize("forward,shift,win", [(True, True, False), (False, True, False), (True, False, True),
                                               (False, True, True), (True, True, True)])
@pytest.mark.parametrize("N,nvec", [(16, 9), (32, 133), (64, 11), (128, 5), (256, 7), (512, 9), (1024, 5), (2048, 3), (4096, 3), (8192, 3)])     # from 32 on: the radix-16 kernels
def test_fft_window_and_shift(gpu, po, forward, shift, win, N, nvec):
    rng = np.random.default_rng(5)
    x = _rc(rng, N * nvec)
    w = np.hamming(N).astype(np.float32) if win else None
    ref = po.fft_vcc(N, forward, w, shift, x)
    blk = gpu.fft_vcc(N, forward, w if win else [], shift)
    assert rel_err_max(blk.work(nvec, x), ref) <= 1e-5
    # set_window: wrong length refused, right length accepted (gr_fft_vcc.cc:55-64)
    assert blk.set_window(np.ones(N - 1, np.float32)) is False
    assert blk.set_window(np.ones(N, np.float32)) is True


@pytest.mark.parametrize("N,nvec", [(32, 300_001), (128, 80_003), (256, 40_003), (512, 19_999), (1024, 9001), (2048, 4001), (4096, 2100), (8192, 1700)])
@pytest.mark.parametrize("forward,shift,win", [(True, False, False), (False, True, False), (True, True, True)])
def test_fft4096_persistent_walk(gpu, po, forward, shift, win, N, nvec):
    """more vectors than resident workgroups (4 per CU plain, 3 windowed; 8192 points: 3 and 2): every workgroup walks
    several vectors (4096: with the next one's points in flight), a ragged last round; every vector against the oracle"""
    rng = np.random.default_rng(11)
    x = _rc(rng, N * nvec)
    w = np.hamming(N).astype(np.float32) if win else None
    ref = po.fft_vcc(N, forward, w, shift, x)
    got = gpu.fft_vcc(N, forward, w if win else [], shift).work(nvec, x)
    err = np.abs(got - ref).reshape(nvec, N).max(1) / np.abs(ref).reshape(nvec, N).max(1)
    assert err.max() <= 1e-5, (int(err.argmax()), float(err.max()))


def test_fft_linearity_and_parseval_full_size(gpu):
    """size-independent properties at the BASELINE size (4096-pt, 4096 vectors = 2^24 samples)"""
    rng = np.random.default_rng(9)
    N, nvec = 4096, 4096
    a = _rc(rng, N * nvec)
    f = gpu.fft_vcc(N, True, [], False)
    A = f.work(nvec, a)
    # Parseval per vector
    ea = (np.abs(a.reshape(nvec, N)) ** 2).sum(1)
    eA = (np.abs(A.reshape(nvec, N)) ** 2).sum(1) / N
    assert np.abs(eA / ea - 1).max() < 1e-5
    # inverse(forward(x)) == N x
    back = gpu.fft_vcc(N, False, [], False).work(nvec, A)
    assert rel_err_max(back / N, a) < 5e-6


def test_fft_errors(gpu):
    with pytest.raises(gpu.GrhipError) as e:
        gpu.fft_vcc(0, True, [], False)
    assert e.value.code == -2                  # gri_fft.cc:104-105: "invalid fft_size"
    with pytest.raises(gpu.GrhipError):
        gpu.fft_vcc((1 << 26) + 2, True, [], False)      # beyond what one handle's work buffers are sized for


def _dft64(x, N, forward, w=None, shift=False):
    """gr_fft_vcc_fftw::work (general/gr_fft_vcc_fftw.cc:64-100) with the transform in float64"""
    v = x.reshape(-1, N).astype(np.complex128)
    if w is not None:
        v = v * w.astype(np.float64)
    elif (not forward) and shift:
        v = np.roll(v, -(N // 2), axis=1)                # dst[i] = in[(i + floor(N/2)) mod N]
    X = np.fft.fft(v, axis=1) if forward else np.fft.ifft(v, axis=1) * N
    if forward and shift:
        X = np.roll(X, -((N + 1) // 2), axis=1)          # out[i] = X[(i + ceil(N/2)) mod N]
    return X.reshape(-1)


# every size class the reference's gri_fft_complex takes (any fft_size > 0, general/gri_fft.cc:97-123) and the
# radix-16 kernels do not: small non-powers of two (direct DFT), other non-powers of two (Bluestein on 128 ... 32768
# points), powers of two above 8192 (four-step), a non-power of two above 4096 (Bluestein on a four-step transform)
@pytest.mark.parametrize("N,nvec", [(3, 50), (7, 33), (48, 21), (100, 13), (127, 5), (129, 7), (1000, 9), (1023, 3), (4097, 2),
                                    (12000, 3), (16384, 5), (32768, 2), (65536, 3), (1 << 20, 1)])
@pytest.mark.parametrize("forward", [True, False])
def test_fft_any_size_vs_float64(gpu, N, nvec, forward):
    rng = np.random.default_rng(N + 3 * int(forward))
    x = _rc(rng, N * nvec)
    ref = _dft64(x, N, forward)
    got = gpu.fft_vcc(N, forward, [], False).work(nvec, x)
    err = np.abs(got - ref).reshape(nvec, N).max(1) / np.abs(ref).reshape(nvec, N).max(1)
    assert err.max() <= 1e-6 * max(np.log2(N), 1), (int(err.argmax()), float(err.max()))


@pytest.mark.parametrize("N,nvec", [(48, 9), (100, 5), (1000, 4), (1001, 3), (16384, 2), (12000, 2)])
@pytest.mark.parametrize("forward,shift,win", [(True, True, False), (False, True, False), (True, False, True),
                                               (False, True, True), (True, True, True)])
def test_fft_any_size_window_and_shift(gpu, N, nvec, forward, shift, win):
    """window, ifft-shift (floor(N/2), backward and only without a window) and fft-shift (ceil(N/2), forward), odd N included"""
    rng = np.random.default_rng(17 + N)
    x = _rc(rng, N * nvec)
    w = np.hamming(N).astype(np.float32) if win else None
    ref = _dft64(x, N, forward, w, shift)
    blk = gpu.fft_vcc(N, forward, w if win else [], shift)
    got = blk.work(nvec, x)
    assert rel_err_max(got, ref) <= 1e-6 * np.log2(N)
    assert blk.set_window(np.ones(N - 1, np.float32)) is False
    assert blk.set_window(np.ones(N, np.float32)) is True


def test_fft_any_size_many_vectors_and_round_trip(gpu):
    """more vectors than one work-buffer chunk holds (Bluestein: 2^25 / L per pass), and inverse(forward(x)) = N x"""
    rng = np.random.default_rng(3)
    N, nvec = 1000, 20000                                # L = 2048: chunks of 16384 vectors
    x = _rc(rng, N * nvec)
    X = gpu.fft_vcc(N, True, [], False).work(nvec, x)
    ref = _dft64(x, N, True)
    err = np.abs(X - ref).reshape(nvec, N).max(1) / np.abs(ref).reshape(nvec, N).max(1)
    assert err.max() <= 1e-5
    back = gpu.fft_vcc(N, False, [], False).work(nvec, X)
    assert rel_err_max(back / N, x) < 5e-6


def _pfb_streams(x, M, tpf):
    return [np.concatenate([np.zeros(tpf, np.complex64), x[j::M]]) for j in range(M)]


@pytest.mark.parametrize("M,ntaps", [(8, 256), (8, 250), (4, 33), (16, 64), (3, 10), (1, 5), (5, 100), (6, 50), (7, 7), (9, 300),
                                     (10, 200), (11, 33), (12, 96), (13, 130), (14, 28), (15, 64), (20, 100), (32, 64),
                                     (32, 1000), (64, 640), (64, 70), (128, 1024), (128, 2500), (256, 512)])
def test_pfb_vs_oracle(gpu, po, M, ntaps):
    rng = np.random.default_rng(M * 1000 + ntaps)
    nout = 1500
    taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
    x = _rc(rng, M * nout)
    o = po.PfbChannelizer(M, taps, 1.0)
    streams = _pfb_streams(x, M, o.taps_per_filter)
    ref, used_ref = o.general_work(nout, streams)
    blk = gpu.pfb_channelizer_ccf(M, taps, 1.0)
    assert blk.history() == o.taps_per_filter + 1
    assert blk.output_multiple() == o.output_multiple
    out0, used0 = blk.general_work(nout, streams)
    assert len(out0) == 0 and used0 == 0          # d_updated from the ctor's set_taps (.cc:169-172)
    out, used = blk.general_work(nout, streams)
    assert used == used_ref == nout
    assert rel_err_max(out, ref) <= 1e-5


@pytest.mark.parametrize("M,tpf,nout", [(8, 32, 512 * 2400 + 77), (8, 8, 512 * 1600 + 1), (8, 48, 512 * 800 + 63),
                                        (4, 32, 512 * 3200 + 5), (10, 20, 512 * 1500 + 9), (5, 33, 512 * 2000 + 311),
                                        (32, 16, 256 * 1500 + 77), (64, 10, 128 * 1100 + 1), (128, 33, 64 * 900 + 63),
                                        (32, 30, 256 * 700 + 5), (64, 40, 128 * 500 + 17), (128, 16, 64 * 1000)])
def test_pfb_persistent_walk(gpu, po, M, tpf, nout):
    """more tiles than resident workgroups: every workgroup walks several tiles (the next tile's samples in flight), and a
    ragged last tile whose vectors past nout fall outside the store descriptor; tpf 32 / 8: taps resident in SGPRs,
    48: taps read in the loop"""
    import torch
    rng = np.random.default_rng(M + tpf)
    taps = rng.uniform(-1, 1, M * tpf).astype(np.float32)
    o = po.PfbChannelizer(M, taps, 1.0)
    assert o.taps_per_filter == tpf
    per = nout + tpf
    ins = np.zeros((M, per), np.complex64)
    ins[:, tpf:] = _rc(rng, M * nout).reshape(M, nout)
    ref, used = o.general_work(nout, [ins[j] for j in range(M)])
    assert used == nout
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d_in = torch.from_numpy(ins.view(np.float32).reshape(M * per, 2)).to(dev)
    d_out = torch.zeros((nout * M + 64, 2), dtype=torch.float32, device=dev)
    pf = gpu.pfb_channelizer_ccf(M, taps, 1.0)
    assert pf.general_work_device(nout, d_in, per, d_out, st) == 0      # d_updated from the ctor's set_taps
    assert pf.general_work_device(nout, d_in, per, d_out, st) == nout
    st.synchronize()
    got = d_out.cpu().numpy().reshape(-1).view(np.complex64)
    assert not got[nout * M:].any()                                     # nothing past the last vector
    err = np.abs(got[:nout * M].reshape(nout, M) - ref).max(1)
    assert err.max() <= 1e-5 * np.abs(ref).max(), (int(err.argmax()), float(err.max()))


@pytest.mark.parametrize("M,os_rate", [(8, 2.0), (8, 4.0), (6, 1.5), (4, 4.0)])
def test_pfb_oversampled(gpu, po, M, os_rate):
    rng = np.random.default_rng(int(M * 10 + os_rate))
    taps = rng.uniform(-1, 1, 5 * M).astype(np.float32)
    o = po.PfbChannelizer(M, taps, os_rate)
    nout = 40 * o.output_multiple * int(os_rate * 2)
    nin = int(round(nout / os_rate))
    x = _rc(rng, M * (nin + 4))
    streams = _pfb_streams(x, M, o.taps_per_filter)
    ref, used_ref = o.general_work(nout, streams)
    blk = gpu.pfb_channelizer_ccf(M, taps, os_rate)
    blk.general_work(nout, streams)
    out, used = blk.general_work(nout, streams)
    assert used == used_ref
    assert rel_err_max(out, ref) <= 1e-5


def test_pfb_errors(gpu):
    with pytest.raises(gpu.GrhipError) as e:
        gpu.pfb_channelizer_ccf(8, np.ones(16, np.float32), 3.0)
    assert e.value.code == -1          # std::invalid_argument


def test_pfb_tone_lands_in_its_channel_full_size(gpu, wl):
    """BASELINE cfg3 size: M=8, 256-tap prototype, 2^21 samples per call x 8 ... a tone
    in channel k comes out in output bin k (property test, size independent)."""
    M, nout = 8, 1 << 18
    taps = wl.lowpass_taps(256, 0.5 / M, 1.0) * 1.0
    blk = gpu.pfb_channelizer_ccf(M, taps, 1.0)
    n = M * nout
    t = np.arange(n)
    for k in (1, 3, 6):
        x = np.exp(2j * np.pi * (k / M + 0.01) * t).astype(np.complex64)
        streams = _pfb_streams(x, M, 32)
        blk.general_work(nout, streams)
        out, _ = blk.general_work(nout, streams)
        p = (np.abs(out[1000:]) ** 2).mean(0)
        assert np.argmax(p) == k and p[k] > 100 * np.delete(p, k).max()


@pytest.mark.parametrize("M,tpf,osr,nout", [(8, 32, 1, 1), (8, 32, 1, 511), (8, 32, 1, 5000), (8, 16, 1, 513), (8, 40, 1, 2000),
                                            (2, 24, 1, 3000), (4, 32, 1, 1025), (16, 8, 1, 1500), (16, 33, 1, 700),
                                            (5, 12, 1, 900), (8, 16, 2, 1200), (32, 16, 1, 600), (12, 10, 3, 999)])
def test_pfb_channelizer_hier_block_in_one_call(gpu, M, tpf, osr, nout):
    """blks2.pfb_channelizer_ccf (blks2impl/pfb_channelizer.py:25-75): ONE interleaved stream in, M streams out, in one
    call -- bit for bit what gr_stream_to_streams -> gr_pfb_channelizer_ccf -> gr_vector_to_streams give (the fused kernel
    at oversample rate 1 and 2 / 4 / 8 / 16 channels; the three kernels behind the same entry for every other shape)"""
    import torch
    rng = np.random.default_rng(M * 100 + tpf)
    taps = rng.uniform(-1, 1, M * tpf - (M // 2)).astype(np.float32)      # ceil(ntaps / M) = tpf, the last filter partly zero
    pf = gpu.pfb_channelizer_ccf(M, taps, osr)
    nout -= nout % pf.output_multiple()
    nout = max(nout, pf.output_multiple())
    assert pf.history() == tpf + 1
    tc = int(np.rint(nout / osr))                                        # items consumed per stream
    x = _rc(rng, tc * M)
    xil = np.concatenate([np.zeros(tpf * M, np.complex64), x])            # taps_per_filter zeros of history on every stream
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d_il = torch.from_numpy(xil.view(np.float32).reshape(-1, 2)).to(dev)
    # the three blocks: stream j = items j, j + M, ... (gr_stream_to_streams.cc:57-63)
    per = tpf + tc + 1
    d_streams = torch.zeros((M, per, 2), dtype=torch.float32, device=dev)
    d_streams[:, :tpf + tc] = d_il.reshape(-1, M, 2).permute(1, 0, 2)
    d_vec = torch.zeros((nout, M, 2), dtype=torch.float32, device=dev)
    assert pf.general_work_device(nout, d_streams, per, d_vec, st) == 0   # d_updated from the ctor's set_taps (.cc:169-172)
    assert pf.general_work_device(nout, d_streams, per, d_vec, st) == nout
    st.synchronize()
    want = d_vec.permute(1, 0, 2).contiguous().cpu().numpy()            # gr_vector_to_streams: channel k = bin k of every vector
    pf2 = gpu.pfb_channelizer_ccf(M, taps, osr)
    stride = nout + 5
    d_out = torch.full((M, stride, 2), 7.0, dtype=torch.float32, device=dev)
    assert pf2.hier_work_device(nout, d_il, d_out, stride, st) == 0
    assert pf2.hier_work_device(nout, d_il, d_out, stride, st) == nout
    st.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:, :nout].view(np.uint32), want.view(np.uint32))
    assert (got[:, nout:] == 7.0).all()                                  # nothing behind the items produced


@pytest.mark.parametrize("ntaps,decim", [(1, 1), (7, 1), (64, 2), (256, 4), (255, 5), (1000, 1), (2049, 3), (300, 16), (100, 8),
                                         (2500, 1)])      # > 2049 taps: the batched overlap-add path
def test_fft_filter_ccc_vs_oracle_and_direct_form(gpu, po, ntaps, decim):
    """gr_fft_filter_ccc (overlap-add): equal to the oracle's restatement of
    gri_fft_filter_ccc_generic within the FFT tolerance (the reference's transforms are FFTW:
    unpinned), equal to the direct-form FIR it stands for, tail carried across calls"""
    rng = np.random.default_rng(ntaps + decim)
    taps = _rc(rng, ntaps)
    ref_blk = po.FftFilterCcc(decim, taps)
    blk = gpu.fft_filter_ccc(decim, taps)
    ns = blk.nsamples()
    assert ns == ref_blk.nsamples and blk.decimation() == decim and blk.history() == 1
    nout = 6 * ns
    x = _rc(rng, nout * decim)
    ref = ref_blk.filter(nout, x)
    # three calls of 1, 2 and 3 output multiples
    got = np.concatenate([blk.work(ns, x[: ns * decim]), blk.work(2 * ns, x[ns * decim: 3 * ns * decim]),
                          blk.work(3 * ns, x[3 * ns * decim:])])
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-6 * np.log2(2 * ns) * scale
    # the function it computes: y[n] = sum_k taps[k] x[n*decim - k] with zeros before the stream
    xin = np.concatenate([np.zeros(ntaps - 1, np.complex64), x])
    direct = po.fir_ccc(taps, xin, nout, decim)
    assert np.abs(got - direct).max() <= 1e-5 * max(np.abs(direct).max(), 1e-3 * np.abs(taps).sum())
    with pytest.raises(gpu.GrhipError):
        blk.work(ns + 1, x)                      # not a multiple of nsamples (the reference asserts)


def test_fft_filter_ccc_set_taps(gpu, po):
    rng = np.random.default_rng(3)
    t1, t2 = _rc(rng, 33), _rc(rng, 200)
    blk = gpu.fft_filter_ccc(1, t1)
    ns1 = blk.nsamples()
    x = _rc(rng, 4096)
    blk.work(ns1, x[:ns1])
    blk.set_taps(t2)
    assert len(blk.work(ns1, x[:ns1])) == 0      # takes effect, produces nothing (gr_fft_filter_ccc.cc:113-118)
    ns2 = blk.nsamples()
    assert ns2 == po.FftFilterCcc(1, t2).nsamples and ns2 != ns1
    got = blk.work(ns2, x[:ns2])                 # tail was cleared by set_taps
    ref = po.FftFilterCcc(1, t2).filter(ns2, x[:ns2])
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("ntaps,decim", [(5000, 1), (8192, 2), (8193, 4)])
def test_fft_filter_ccc_beyond_4096_taps(gpu, po, ntaps, decim):
    """filters whose transform is larger than the register kernels take (fftsize = 2 * 2^ceil(log2 ntaps) >= 16384,
    gri_fft_filter_ccc_generic.cc:98-118 sizes it from ntaps without a cap): the four-step form of the same
    overlap-add; sizes as the reference computes them, results against the direct-form FIR the block stands for
    (the oracle's O(N^2) double transform is not run at these sizes), tail carried across calls"""
    rng = np.random.default_rng(ntaps)
    taps = (_rc(rng, ntaps) / np.sqrt(ntaps)).astype(np.complex64)
    blk = gpu.fft_filter_ccc(decim, taps)
    fftsize = int(2 * 2 ** np.ceil(np.log2(ntaps)))
    ns = blk.nsamples()
    assert ns == fftsize - ntaps + 1 and blk.decimation() == decim
    nout = 3 * ns
    x = _rc(rng, nout * decim)
    got = np.concatenate([blk.work(ns, x[: ns * decim]), blk.work(2 * ns, x[ns * decim:])])
    xin = np.concatenate([np.zeros(ntaps - 1, np.complex64), x])
    direct = po.fir_ccc(taps, xin, nout, decim)
    assert np.abs(got - direct).max() <= 1e-5 * max(np.abs(direct).max(), 1e-3 * np.abs(taps).sum())
