"""-m gpu parity tests of the FIR family through the C ABI (ctypes) against the
CPU oracle.  Tolerances: GENERIC mode bit-exact; FAST mode 1e-5 relative
(BASELINE.json north_star), measured as ||y-ref||inf / ||ref||inf and, where
|ref| is not tiny, per element (filter/qa_gr_fir_ccf.cc:151-152 style)."""
import numpy as np
import pytest

from conftest import bits_equal, demod_close, demod_report, rel_err_max

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rand_c(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


@pytest.mark.parametrize("kind", ["fff", "ccf", "ccc"])
@pytest.mark.parametrize("ntaps", [1, 2, 7, 8, 64, 255, 256])
@pytest.mark.parametrize("decim", [1, 4])
def test_fir_generic_mode_bit_exact(gpu, po, kind, ntaps, decim):
    rng = np.random.default_rng(ntaps * 10 + decim)
    n = 3001
    nin = n * decim + ntaps - 1          # what the scheduler guarantees (gr_sync_decimator.cc:46-50)
    if kind == "fff":
        x = rng.uniform(-1, 1, nin).astype(np.float32)
        taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
        blk = gpu.fir_filter_fff(decim, taps)
        ref = po.fir_fff(taps, x, n, decim)
    elif kind == "ccf":
        x = _rand_c(rng, nin)
        taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
        blk = gpu.fir_filter_ccf(decim, taps)
        ref = po.fir_ccf(taps, x, n, decim)
    else:
        x = _rand_c(rng, nin)
        taps = _rand_c(rng, ntaps)
        blk = gpu.fir_filter_ccc(decim, taps)
        ref = po.fir_ccc(taps, x, n, decim)
    blk.set_mode(gpu.MODE_GENERIC)
    assert blk.history() == ntaps
    got = blk.work(n, x)
    assert bits_equal(got, ref)


@pytest.mark.parametrize("kind", ["ccf", "ccc"])
@pytest.mark.parametrize("decim", [1, 2, 4, 3])
@pytest.mark.parametrize("ntaps", [8, 9, 15, 16, 17, 33, 100, 257])
def test_fir_generic_window_kernel_shapes(gpu, po, kind, ntaps, decim):
    """the bit-exact mode's tiled kernels (decimation 1 / 2 / 4: samples in register windows, taps in blocks of 4 D with a
    guarded last block; 3: samples from LDS): tap counts around the block sizes, an output count that is not a multiple of the
    tile, exact and negative zeros among samples and taps (a padding tap would turn a -0 sum into +0)"""
    rng = np.random.default_rng(ntaps * 100 + decim)
    n = 5003
    nin = n * decim + ntaps - 1
    x = _rand_c(rng, nin)
    x[rng.integers(0, nin, 200)] = 0
    x[rng.integers(0, nin, 200)] = np.complex64(complex(-0.0, -0.0))
    x[1000:1000 + 2 * ntaps * decim] = np.complex64(complex(-0.0, 0.0))      # whole outputs that are sums of signed zeros
    if kind == "ccf":
        taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
        taps[::5] = -0.0
        blk = gpu.fir_filter_ccf(decim, taps)
        ref = po.fir_ccf(taps, x, n, decim)
    else:
        taps = _rand_c(rng, ntaps)
        taps[::5] = np.complex64(complex(-0.0, 0.0))
        blk = gpu.fir_filter_ccc(decim, taps)
        ref = po.fir_ccc(taps, x, n, decim)
    blk.set_mode(gpu.MODE_GENERIC)
    got = blk.work(n, x)
    assert bits_equal(got, ref)


@pytest.mark.parametrize("kind", ["ccf", "ccc"])
@pytest.mark.parametrize("ntaps,decim", [(1, 1), (3, 1), (64, 1), (65, 2), (256, 4), (255, 4), (256, 1), (31, 8), (40, 3),
                                         (200, 3), (1500, 1), (2049, 5), (600, 8), (700, 16), (1100, 2),
                                         (900, 4), (400, 20), (100, 5), (64, 7),
                                         (2000, 50)])    # from (200, 3) on: overlap-save engine (folded inverse at D = 2..16), last four (and
                                                         # (31, 8), (40, 3)): high-decimation direct kernel
def test_fir_fast_mode_tolerance(gpu, po, kind, ntaps, decim):
    rng = np.random.default_rng(1000 + ntaps * 10 + decim)
    n = 5003           # not a multiple of the tile: exercises the ragged tail
    nin = n * decim + ntaps - 1          # what the scheduler guarantees (gr_sync_decimator.cc:46-50)
    x = _rand_c(rng, nin)
    if kind == "ccf":
        taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
        blk = gpu.fir_filter_ccf(decim, taps)
        ref = po.fir_ccf(taps, x, n, decim)
        bound = np.abs(taps).sum()
    else:
        taps = _rand_c(rng, ntaps)
        blk = gpu.fir_filter_ccc(decim, taps)
        ref = po.fir_ccc(taps, x, n, decim)
        bound = np.abs(taps).sum() * np.sqrt(2)
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(n, x)
    assert got.shape == ref.shape
    # random taps: outputs have heavy cancellation, so bound the error by the
    # worst-case output magnitude ||taps||_1 * ||x||_inf
    assert np.abs(got - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3 * bound)


@pytest.mark.parametrize("kind,ntaps,decim,n", [("ccc", 1100, 1, 3_300_000), ("ccc", 600, 4, 1_400_000), ("ccc", 200, 3, 1_500_000),
                                                ("fff", 1500, 1, 4_000_000), ("fff", 500, 8, 700_000)])
def test_overlap_save_engine_persistent_walk(gpu, po, kind, ntaps, decim, n):
    """more 4096-point blocks than resident workgroups (three per CU): every workgroup walks several blocks with the next
    block's points in flight; full-size inverse with decimation 1 (descriptor stores) and 3 (generic), folded inverse at
    4 and 8, complex and real data; every output against the oracle's direct form"""
    rng = np.random.default_rng(ntaps + decim)
    nin = n * decim + ntaps - 1
    if kind == "ccc":
        x = _rand_c(rng, nin)
        taps = _rand_c(rng, ntaps) / np.float32(ntaps ** 0.5)
        blk, ref = gpu.fir_filter_ccc(decim, taps), po.fir_ccc(taps, x, n, decim)
    else:
        x = rng.uniform(-1, 1, nin).astype(np.float32)
        taps = (rng.uniform(-1, 1, ntaps) / ntaps ** 0.5).astype(np.float32)
        blk, ref = gpu.fir_filter_fff(decim, taps), po.fir_fff(taps, x, n, decim)
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(n, x)
    err = np.abs(got - ref)
    assert err.max() <= TOL * np.abs(ref).max(), (int(err.argmax()), float(err.max()), float(np.abs(ref).max()))


def test_overlap_save_engine_call_longer_than_2GB(gpu, po):
    """one device call of 2.15 GB of input: the engine addresses a launch through 32-bit byte offsets, so the call goes out
    in two pieces of whole blocks; outputs around the seam, at the very start and at the very end against the direct form"""
    import torch
    ntaps, L = 64, 4096 - 63
    seam = ((0x7fff0000 // 8 - 2 * 4096) // L) * L               # first output of the second piece
    n = seam + 3 * L + 77
    nin = n + ntaps - 1
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    d_x = torch.empty((nin, 2), dtype=torch.float32, device=dev)
    d_x.uniform_(-1, 1, generator=gen)
    d_y = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    taps = np.random.default_rng(6).uniform(-1, 1, ntaps).astype(np.float32)
    blk = gpu.fir_filter_ccf(1, taps)
    blk.set_mode(gpu.MODE_FAST_VALU)                            # (64 taps at decimation 1: the overlap-save engine)
    assert blk.work_device(n, d_x, d_y, st) == n
    st.synchronize()
    for a in (0, seam - 5000, n - 5000):
        m = 5000 if a + 5000 <= n else n - a
        xs = d_x[a:a + m + ntaps - 1].cpu().numpy().reshape(-1).view(np.complex64)
        ref = po.fir_ccf(taps, xs, m, 1)
        got = d_y[a:a + m].cpu().numpy().reshape(-1).view(np.complex64)
        assert np.abs(got - ref).max() <= TOL * np.abs(ref).max(), (a, float(np.abs(got - ref).max()))
    del d_x, d_y
    torch.cuda.empty_cache()


@pytest.mark.parametrize("decim,ntaps", [(10, 200), (20, 400), (8, 64)])
def test_fused_demod_other_decimations_stream_and_switch_modes(gpu, po, wl, decim, ntaps):
    """the fused xlating -> demodulator block at decimations the tiled / matrix-core kernels do not take: the direct kernel's
    own demodulator epilogue (no rotator table), in calls of odd sizes with the one-sample carry between them, switching to
    the bit-exact mode and back in mid-stream (the carry changes frame)"""
    c = wl.CFG2
    n = 1_200_000 // decim * decim
    x = wl.fsk4_capture(n, stream_id=4)
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], 3.0, x)
    nout = n // decim
    hist = ntaps - 1
    xin = wl.with_history(x, hist)
    blk = gpu.xlating_demod(decim, proto, c["center_freq"], c["fs"], 3.0)
    pieces = [1, 2, 255, 256, 257, 511, 513, 10_001, 1, 33_333]
    modes = [gpu.MODE_FAST, gpu.MODE_FAST, gpu.MODE_GENERIC, gpu.MODE_FAST, gpu.MODE_FAST_VALU]
    out, pos, k = [], 0, 0
    while pos < nout:
        m = min(pieces[k % len(pieces)], nout - pos)
        blk.set_mode(modes[k % len(modes)])
        out.append(blk.work(m, xin[pos * decim: pos * decim + m * decim + hist]))
        pos += m
        k += 1
    got = np.concatenate(out)
    ok, worst = demod_close(got, ref, skip=max(64, ntaps // decim + 1), gain=3.0)
    assert ok, worst


def test_fir_integer_data_exact_all_modes(gpu, po):
    """integer-valued data: every summation order is exact
    (filter/qa_gr_fir_ccf.cc:103-159 uses the same trick)"""
    rng = np.random.default_rng(7)
    for ntaps in range(0, 10):
        for n in (1, 2, 17):
            x = (rng.integers(-8, 8, (n + ntaps, 2))).astype(np.float32).view(np.complex64).reshape(-1)
            taps = rng.integers(-8, 8, ntaps).astype(np.float32)
            ref = po.fir_ccf(taps, x, n, 1)
            for mode in (gpu.MODE_FAST, gpu.MODE_GENERIC):
                blk = gpu.fir_filter_ccf(1, taps)
                blk.set_mode(mode)
                got = blk.work(n, x)
                assert np.array_equal(got, ref), (ntaps, n, mode)


def test_fir_fff_known_vectors(gpu):
    """filter/qa_gr_fir_fff.cc:58-76"""
    inp = np.array([234, -4, 23, -56, 45, 98, -23, -7], np.float32)
    blk = gpu.fir_filter_fff(1, [-3])
    assert np.array_equal(blk.work(8, inp), np.array([-702, 12, -69, 168, -135, -294, 69, 21], np.float32))
    blk = gpu.fir_filter_fff(1, [-4, 5])
    assert np.array_equal(blk.work(7, inp), np.array([1186, -112, 339, -460, -167, 582, -87], np.float32))
    # filterN / filterNdec seam
    assert np.array_equal(blk.filterNdec(inp, 4, 2), np.array([1186, 339, -167, -87], np.float32))


def test_fir_set_taps_returns_zero_once(gpu, po):
    """filter/gr_fir_filter_XXX.cc.t:74-79"""
    rng = np.random.default_rng(3)
    x = _rand_c(rng, 2000)
    blk = gpu.fir_filter_ccf(1, [1.0, 2.0])
    blk.set_mode(gpu.MODE_GENERIC)
    y0 = blk.work(100, x)
    assert len(y0) == 100
    blk.set_taps([0.5, 0.25, 0.125])
    assert len(blk.work(100, x)) == 0          # update consumed, nothing produced
    assert blk.history() == 3
    y1 = blk.work(100, x)
    assert bits_equal(y1, po.fir_ccf([0.5, 0.25, 0.125], x, 100, 1))


def test_fir_through_scheduler_shim_cfg1(gpu, po, wl):
    """config 1: gr_fir_filter_ccf 64-tap low-pass on a complex vector through the
    chunked sync-block runner (history zeros, 4096-item calls)."""
    n = 200_000
    x = wl.uniform_complex(n)
    taps = wl.lowpass_taps(64, 0.1, 1.0)
    blk = gpu.fir_filter_ccf(1, taps)
    got = gpu.run_sync_block(blk, x, chunk=4096)
    ref = po.fir_ccf(taps, wl.with_history(x, 63), n, 1)
    assert rel_err_max(got, ref) <= TOL
    blk2 = gpu.fir_filter_ccf(1, taps)
    blk2.set_mode(gpu.MODE_GENERIC)
    assert bits_equal(gpu.run_sync_block(blk2, x, chunk=4096), ref)


# ---------------------------------------------------------------------------
# freq_xlating_fir_filter_ccc
# ---------------------------------------------------------------------------
def test_xlating_generic_bit_exact_and_rotator_carry(gpu, po, wl):
    c = wl.CFG2
    n = 120_000
    x = wl.fsk4_capture(n)
    proto = wl.cfg2_proto_taps()
    nout = n // c["decim"]
    xin = wl.with_history(x, len(proto) - 1)
    ref = po.Xlating(c["decim"], proto, c["center_freq"], c["fs"]).work(xin, nout)
    blk = gpu.freq_xlating_fir_filter_ccc(c["decim"], proto, c["center_freq"], c["fs"])
    blk.set_mode(gpu.MODE_GENERIC)
    assert blk.history() == 256
    # one big call
    assert bits_equal(blk.work(nout, xin), ref)
    # chunked calls: rotator phase/counter carried across calls (gr_rotator.h:40-50,
    # renormalisation every 512 outputs crosses chunk boundaries)
    blk.reset()
    got = gpu.run_sync_block(blk, x, chunk=1000)
    assert bits_equal(got, ref)


@pytest.mark.parametrize("complex_proto", [False, True])
@pytest.mark.parametrize("decim,ntaps", [(10, 256), (5, 1200), (25, 400), (8, 300), (16, 512), (20, 400), (7, 64), (3, 40)])
def test_xlating_fast_other_decimations(gpu, po, wl, decim, ntaps, complex_proto):
    """decimations the tiled kernel does not take (and prototypes beyond its 1024 taps): FAST mode goes through the
    high-decimation direct kernel (pre-mix form for a real prototype, complex taps otherwise) or the overlap-save
    engine, + rotator table (+ stand-alone demodulator), chunked calls included"""
    c = wl.CFG2
    n = 300_000
    x = wl.fsk4_capture(n, stream_id=31)
    proto = wl.lowpass_taps(ntaps, 0.02, 1.0).astype(np.complex64)
    if complex_proto:
        proto = (proto * np.exp(1j * 0.013 * np.arange(ntaps))).astype(np.complex64)
    nout = n // decim
    xin = wl.with_history(x[: nout * decim], ntaps - 1)
    ref = po.Xlating(decim, proto, c["center_freq"], c["fs"]).work(xin, nout)
    blk = gpu.freq_xlating_fir_filter_ccc(decim, proto, c["center_freq"], c["fs"])
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(nout, xin)
    assert rel_err_max(got, ref) <= TOL
    blk.reset()
    got2 = gpu.run_sync_block(blk, x[: nout * decim], chunk=1000)
    assert rel_err_max(got2, ref) <= TOL
    # exactly the items the scheduler guarantees ((n-1)*decim + ntaps) followed by NaNs: nothing past them is used
    blk.reset()
    guarded = np.concatenate([xin[: (nout - 1) * decim + ntaps], np.full(decim + 8, np.nan + 1j * np.nan, np.complex64)])
    got3 = blk.work(nout, guarded)
    assert np.isfinite(got3).all() and rel_err_max(got3, ref) <= TOL
    # fused handle: same path + demodulator, carry across calls
    gain = 2.0
    dref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], gain, x[: nout * decim])
    fz = gpu.xlating_demod(decim, proto, c["center_freq"], c["fs"], gain)
    d = gpu.run_sync_block(fz, x[: nout * decim], chunk=7000)
    ok, worst = demod_close(d, dref, skip=max(64, ntaps // decim + 1), gain=gain)
    assert ok, worst


@pytest.mark.parametrize("complex_proto", [False, True])
def test_xlating_fast_tolerance(gpu, po, wl, complex_proto):
    c = wl.CFG2
    n = 400_000
    x = wl.fsk4_capture(n, stream_id=1)
    proto = wl.cfg2_proto_taps()
    if complex_proto:   # forces the complex-tap tiled path (no pre-mix)
        proto = (proto * np.exp(1j * 0.01 * np.arange(len(proto)))).astype(np.complex64)
    nout = n // c["decim"]
    xin = wl.with_history(x, len(proto) - 1)
    ref = po.Xlating(c["decim"], proto, c["center_freq"], c["fs"]).work(xin, nout)
    blk = gpu.freq_xlating_fir_filter_ccc(c["decim"], proto, c["center_freq"], c["fs"])
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(nout, xin)
    assert rel_err_max(got, ref) <= TOL
    # per element where the reference is not tiny (pass-band signal, |y| ~ 1)
    m = np.abs(ref) > 0.1 * np.abs(ref).max()
    assert (np.abs(got - ref)[m] / np.abs(ref)[m]).max() <= TOL
    # chunked: same stream in 4096-output calls must continue seamlessly
    blk.reset()
    got2 = gpu.run_sync_block(blk, x, chunk=4096)
    assert rel_err_max(got2, ref) <= TOL


def test_xlating_set_center_freq_keeps_phase(gpu, po, wl):
    """set_center_freq -> next work returns 0, then new composite taps and
    increment with the rotator phase carried on (.cc.t:86-114)"""
    c = wl.CFG2
    n = 40_000
    x = wl.fsk4_capture(n, stream_id=2)
    proto = wl.cfg2_proto_taps()
    blk = gpu.freq_xlating_fir_filter_ccc(4, proto, c["center_freq"], c["fs"])
    blk.set_mode(gpu.MODE_GENERIC)
    xin = wl.with_history(x, 255)
    a = blk.work(2000, xin)
    blk.set_center_freq(1.0e6)
    assert len(blk.work(10, xin)) == 0
    b = blk.work(3000, xin[8000:])
    # oracle: same sequence by hand
    o = po.Xlating(4, proto, c["center_freq"], c["fs"])
    ra = o.work(xin, 2000)
    ph, inc, cnt = o.rot()
    o2 = po.Xlating(4, proto, 1.0e6, c["fs"])
    ph2, inc2, _ = o2.rot()
    phases = np.empty(3000, np.complex64)
    pr, pi = np.float32(ph.real), np.float32(ph.imag)
    ir, ii = np.float32(inc2.real), np.float32(inc2.imag)
    k = cnt
    for i in range(3000):
        phases[i] = complex(pr, pi)
        k += 1
        ac, bd, ad, bc = np.float32(pr * ir), np.float32(pi * ii), np.float32(pr * ii), np.float32(pi * ir)
        pr, pi = np.float32(ac - bd), np.float32(ad + bc)
        if k % 512 == 0:
            a_ = np.float32(np.hypot(np.float64(pr), np.float64(pi)))
            pr, pi = np.float32(pr / a_), np.float32(pi / a_)
    fir = po.fir_ccc(o2.ctaps()[::-1].copy(), xin[8000:], 3000, 4)
    fr, fi = fir.real.astype(np.float32), fir.imag.astype(np.float32)
    cr, ci = phases.real.astype(np.float32), phases.imag.astype(np.float32)
    rb = ((fr * cr).astype(np.float32) - (fi * ci).astype(np.float32)) + 1j * ((fr * ci).astype(np.float32) + (fi * cr).astype(np.float32))
    assert bits_equal(a, ra)
    assert rel_err_max(b, rb.astype(np.complex64)) <= 2e-7   # hypot rounding path differs in numpy; ~1 ulp


# ---------------------------------------------------------------------------
# quadrature_demod_cf and the fused hier block
# ---------------------------------------------------------------------------
def test_quad_demod_bit_exact(gpu, po):
    rng = np.random.default_rng(11)
    n = 100_003
    x = _rand_c(rng, n + 1)
    # axes, zeros, equal magnitudes: all octant/shortcut branches of gr_fast_atan2f
    x[10] = 0; x[11] = 0; x[20] = 1; x[21] = 1j; x[22] = -1; x[23] = -1j; x[24] = 1 + 1j; x[25] = 1e-4 + 1j
    blk = gpu.quadrature_demod_cf(4.9)
    got = blk.work(n, x)
    ref = po.quad_demod_cf(4.9, x, n)
    assert bits_equal(got, ref)


def test_fused_xlating_demod_cfg2(gpu, po, wl):
    c = wl.CFG2
    n = 1_000_000
    x = wl.fsk4_capture(n, stream_id=3)
    proto = wl.cfg2_proto_taps()
    nout = n // c["decim"]
    ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    xin = wl.with_history(x, 255)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    got = blk.work(nout, xin)
    ok, worst = demod_close(got, ref)
    assert ok, worst
    # generic mode: FIR+rotator bit exact, demod bit exact => whole chain bit exact
    blk.reset(); blk.set_mode(gpu.MODE_GENERIC)
    assert bits_equal(blk.work(nout, xin), ref)
    # chunked fast: demodulator's previous sample carried across calls
    blk.reset(); blk.set_mode(gpu.MODE_FAST)
    got2 = gpu.run_sync_block(blk, x, chunk=8192)
    ok, worst = demod_close(got2, ref)
    assert ok, worst


@pytest.mark.parametrize("decim,ntaps", [(4, 256), (2, 100), (1, 33), (4, 17)])
def test_generic_xlating_demod_in_one_kernel(gpu, po, wl, decim, ntaps):
    """the bit-exact mode's fused kernel (gr_fir_ccc_generic order + rotator + gr_quadrature_demod_cf, decimation 1 / 2 / 4):
    one call, then calls of uneven sizes -- the demodulator's previous sample and the rotator's phase carried across calls,
    sizes below two tiles (where the FIR and the demodulator run as two kernels) mixed in, tile edges (1020 new outputs per
    tile) among the sizes -- all bit-exact against the oracle's chain"""
    c = wl.CFG2
    n = 200_000 * decim
    x = wl.fsk4_capture(n, stream_id=17)
    proto = wl.cfg2_proto_taps() if ntaps == 256 else wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    nout = n // decim
    ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    blk = gpu.xlating_demod(decim, proto, c["center_freq"], c["fs"], c["demod_gain"])
    blk.set_mode(gpu.MODE_GENERIC)
    xin = wl.with_history(x, ntaps - 1)
    assert bits_equal(blk.work(nout, xin), ref)
    blk.reset()
    got, pos = [], 0
    for m in (2048, 1020, 2040, 2041, 3061, 500, 4081, 10_000, 2049, 65_536):
        m = min(m, nout - pos)
        got.append(blk.work(m, xin[pos * decim: (pos + m) * decim + ntaps - 1]))
        pos += m
    got.append(blk.work(nout - pos, xin[pos * decim:]))
    assert bits_equal(np.concatenate(got), ref)


@pytest.mark.parametrize("stride_pad,n", [(0, 2_000_000), (63, 1_900_001)])
def test_run_captures_batched_vs_oracle(gpu, po, wl, stride_pad, n):
    """one batched launch over 5 captures: more tiles than the static share of the
    persistent grid (the tile queue hands out the rest), odd stream stride (the streams'
    16-byte alignment parity alternates, so the pre-mix phasors are rebuilt between
    tiles), capture length not a multiple of the decimation or the tile"""
    import torch
    c = wl.CFG2
    S = 5
    proto = wl.cfg2_proto_taps()
    xs = [wl.fsk4_capture(n, stream_id=70 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + stride_pad
    d_in = torch.zeros((S * stride + 8, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s * stride: s * stride + n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    nout = n // c["decim"]
    ostride = ((nout + 3) // 4) * 4
    d_out = torch.zeros((S, ostride), dtype=torch.float32, device=dev)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    st = torch.cuda.Stream(device=dev)
    for rep in range(2):            # second launch: the queue re-armed itself
        d_out.zero_()
        blk.run_captures_device(S, n, d_in, stride, d_out, ostride, st)
        st.synchronize()
        got = d_out.cpu().numpy()
        for s in range(S):
            ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"],
                                         xs[s][: nout * c["decim"]])
            ok, worst = demod_close(got[s, :nout], ref)
            assert ok, (rep, s, worst)


@pytest.mark.parametrize("decim,ntaps,stride_pad,n", [(4, 256, 0, 300_000), (4, 256, 62, 250_003), (2, 100, 2, 120_001),
                                                      (1, 32, 0, 40_000), (1, 33, 4, 30_001)])
def test_run_captures_generic_mode_bit_exact(gpu, po, wl, decim, ntaps, stride_pad, n):
    """the multi-capture entry in the bit-exact mode: every capture of a batch through the fused generic-order kernel in one
    launch (fresh history synthesised, never read: the captures' rows start 8 bytes off a 16-byte boundary once the history
    is counted in when ntaps is even), each bit-exact against the oracle's chain; even strides, lengths that are no
    multiple of the decimation"""
    import torch
    c = wl.CFG2
    S = 3
    proto = wl.cfg2_proto_taps() if ntaps == 256 else wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    xs = [wl.fsk4_capture(n, stream_id=110 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + stride_pad + (n + stride_pad) % 2
    d_in = torch.zeros((S * stride + 8, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s * stride: s * stride + n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    nout = n // decim
    ostride = ((nout + 3) // 4) * 4
    d_out = torch.zeros((S, ostride), dtype=torch.float32, device=dev)
    blk = gpu.xlating_demod(decim, proto, c["center_freq"], c["fs"], c["demod_gain"])
    blk.set_mode(gpu.MODE_GENERIC)
    st = torch.cuda.Stream(device=dev)
    blk.run_captures_device(S, n, d_in, stride, d_out, ostride, st)
    st.synchronize()
    got = d_out.cpu().numpy()
    for s in range(S):
        ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], c["demod_gain"], xs[s][: nout * decim])
        assert bits_equal(got[s, :nout], ref), s


@pytest.mark.parametrize("decim,ntaps,stride_pad,n", [(20, 400, 0, 1_000_000), (5, 200, 63, 700_003), (3, 96, 1, 300_001),
                                                      (8, 256, 7, 500_000)])
def test_run_captures_other_decimations(gpu, po, wl, decim, ntaps, stride_pad, n):
    """the batched launch at decimations only the direct kernel takes: (tile, stream) pairs walked by persistent workgroups,
    the captures' zero history by range check, odd strides, lengths that are no multiple of the decimation"""
    import torch
    c = wl.CFG2
    S = 3
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    xs = [wl.fsk4_capture(n, stream_id=90 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + stride_pad
    d_in = torch.zeros((S * stride + 8, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s * stride: s * stride + n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    nout = n // decim
    ostride = nout + 3
    d_out = torch.zeros((S, ostride), dtype=torch.float32, device=dev)
    blk = gpu.xlating_demod(decim, proto, c["center_freq"], c["fs"], 3.0)
    st = torch.cuda.Stream(device=dev)
    for mode in (gpu.MODE_FAST, gpu.MODE_FAST_VALU):
        blk.set_mode(mode)
        d_out.zero_()
        blk.run_captures_device(S, n, d_in, stride, d_out, ostride, st)
        st.synchronize()
        got = d_out.cpu().numpy()
        for s in range(S):
            ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], 3.0, xs[s][: nout * decim])
            ok, worst = demod_close(got[s, :nout], ref, gain=3.0)
            assert ok, (mode, s, worst)
            assert not got[s, nout:].any()


@pytest.mark.parametrize("nout", [1, 5, 8, 9, 2015, 2016, 2017, 4033])
def test_fused_xlating_demod_small_and_tile_edges(gpu, po, wl, nout):
    """output counts around the lane (8) and tile (2016 new outputs) granularity, two calls each
    so that the one-sample carry crosses a call at every size"""
    c = wl.CFG2
    n = 2 * nout * c["decim"]
    x = wl.fsk4_capture(max(n, 4096), stream_id=20)[:n]
    proto = wl.cfg2_proto_taps()
    ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    xin = wl.with_history(x, 255)
    a = blk.work(nout, xin[: nout * 4 + 255])
    b = blk.work(nout, xin[nout * 4: 2 * nout * 4 + 255])
    got = np.concatenate([a, b])
    ok, worst = demod_close(got, ref[: 2 * nout])
    assert ok, worst


def test_fused_xlating_demod_mode_switch_mid_stream(gpu, po, wl):
    """FAST and GENERIC keep the demodulator's one-sample carry in different frames
    (fir_kernels.h, EPI_DEMOD); switching between work() calls converts it"""
    c = wl.CFG2
    n = 120_000
    x = wl.fsk4_capture(n, stream_id=9)
    proto = wl.cfg2_proto_taps()
    ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    xin = wl.with_history(x, 255)
    out, pos = [], 0
    for k, nchunk in enumerate((10_000, 5_000, 7_500, 7_500)):
        blk.set_mode(gpu.MODE_FAST if k % 2 == 0 else gpu.MODE_GENERIC)
        out.append(blk.work(nchunk, xin[pos * 4: (pos + nchunk) * 4 + 255]))
        pos += nchunk
    got = np.concatenate(out)
    ok, worst = demod_close(got, ref[:pos])
    assert ok, worst


def test_fused_xlating_demod_complex_prototype(gpu, po, wl):
    """complex prototype taps: no pre-mix form, the tiled kernel writes y and the
    stand-alone demodulator follows (chunked, so the carry crosses calls)"""
    c = wl.CFG2
    n = 300_000
    x = wl.fsk4_capture(n, stream_id=11)
    rng = np.random.default_rng(5)
    proto = (wl.cfg2_proto_taps().real * np.exp(1j * 0.01 * np.arange(256))).astype(np.complex64)
    ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    got = gpu.run_sync_block(blk, x, chunk=50_000)
    ok, worst = demod_close(got, ref)
    assert ok, worst


def test_unfused_pipeline_equals_fused(gpu, po, wl):
    """tb.connect(xlating, demod) as two blocks vs the fused hier block"""
    c = wl.CFG2
    n = 200_000
    x = wl.fsk4_capture(n, stream_id=4)
    proto = wl.cfg2_proto_taps()
    nout = n // 4
    xl = gpu.freq_xlating_fir_filter_ccc(4, proto, c["center_freq"], c["fs"])
    qd = gpu.quadrature_demod_cf(c["demod_gain"])
    y = xl.work(nout, wl.with_history(x, 255))
    d = qd.work(nout, wl.with_history(y, 1))
    fused = gpu.xlating_demod(4, proto, c["center_freq"], c["fs"], c["demod_gain"]).work(nout, wl.with_history(x, 255))
    # only the tile-boundary predecessor differs (tree-order sum): well inside tolerance
    ok, worst = demod_close(d, fused, tol=4e-6)
    assert ok, worst


def test_errors_and_edges(gpu):
    with pytest.raises(gpu.GrhipError):
        gpu.fir_filter_ccf(0, [1.0])
    blk = gpu.fir_filter_ccf(1, [])
    assert blk.history() == 1
    assert len(blk.work(0, np.zeros(0, np.complex64))) == 0
    out = blk.work(5, np.ones(5, np.complex64))          # zero taps -> zeros
    assert np.array_equal(out, np.zeros(5, np.complex64))


@pytest.mark.parametrize("ntaps,decim,n", [(1, 1, 9), (3, 1, 4097), (256, 1, 100_001), (255, 2, 50_000), (64, 2, 33_333),
                                           (17, 1, 2), (300, 4, 5000), (1500, 1, 20_001), (100, 3, 777),
                                           (2049, 7, 3000), (500, 8, 2000), (500, 16, 999)])      # from (300, 4) on: real-data overlap-save engine
def test_fir_fff_fast_mode(gpu, po, ntaps, decim, n):
    """gr_fir_filter_fff through the tiled kernel's float-pair mode (decimation 1, 2), the real-data
    overlap-save engine (other decimations, more than 1024 taps) or the generic-order kernel (short
    filters at other decimations), odd output counts included"""
    rng = np.random.default_rng(ntaps * 7 + decim)
    nin = n * decim + ntaps - 1
    x = rng.uniform(-1, 1, nin).astype(np.float32)
    taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
    blk = gpu.fir_filter_fff(decim, taps)
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(n, x)
    ref = po.fir_fff(taps, x, n, decim)
    bound = np.abs(taps).sum()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL * max(np.abs(ref).max(), 1e-3 * bound)
    # integer-valued data: exact in any order for the direct form (the overlap-save engine, used where the
    # float-pair mode does not reach, rounds in its transforms)
    xi = rng.integers(-8, 8, nin).astype(np.float32)
    ti = rng.integers(-4, 4, ntaps).astype(np.float32)
    blk2 = gpu.fir_filter_fff(decim, ti)
    engine = ntaps >= 48 and (decim > 2 or ntaps > 1024 or ntaps // decim > 176)
    gi, ri = blk2.work(n, xi), po.fir_fff(ti, xi, n, decim)
    if engine:
        assert np.abs(gi - ri).max() <= TOL * np.abs(ri).max()
    else:
        assert np.array_equal(gi, ri)


def test_fir_random_shapes_fast_mode(gpu, po):
    """a seeded sweep over kind / tap count / decimation / output count / call chunking:
    tap counts around the step (8) and triple-step (24) granularity of the MAC loop, above the
    tiled kernel's 1024-tap limit (generic-order fallback), output counts around the tile"""
    rng = np.random.default_rng(20261004)
    kinds = ["ccf", "ccc", "fff"]
    for case in range(48):
        kind = kinds[case % 3]
        ntaps = int(rng.choice([1, 2, 7, 8, 9, 23, 24, 25, 63, 64, 65, 191, 192, 193, 257, 500, 1023, 1025, 1500]))
        decim = int(rng.choice([1, 1, 2, 3, 4]))
        n = int(rng.choice([1, 8, 9, 511, 2015, 2016, 2017, 2048, 4097, 12345]))
        nin = n * decim + ntaps - 1
        if kind == "fff":
            x = rng.uniform(-1, 1, nin).astype(np.float32)
            taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
            blk = gpu.fir_filter_fff(decim, taps)
            ref = po.fir_fff(taps, x, n, decim)
            bound = np.abs(taps).sum()
        elif kind == "ccf":
            x = _rand_c(rng, nin)
            taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
            blk = gpu.fir_filter_ccf(decim, taps)
            ref = po.fir_ccf(taps, x, n, decim)
            bound = np.abs(taps).sum()
        else:
            x = _rand_c(rng, nin)
            taps = _rand_c(rng, ntaps)
            blk = gpu.fir_filter_ccc(decim, taps)
            ref = po.fir_ccc(taps, x, n, decim)
            bound = np.abs(taps).sum() * np.sqrt(2)
        blk.set_mode(gpu.MODE_FAST)
        # two calls when possible: the second starts in the middle of the stream
        n1 = n // 2
        parts = []
        if n1 > 0:
            parts.append(blk.work(n1, x[: n1 * decim + ntaps - 1]))
        parts.append(blk.work(n - n1, x[n1 * decim:]))
        got = np.concatenate(parts)
        assert got.shape == ref.shape, (kind, ntaps, decim, n)
        err = np.abs(got - ref).max()
        assert err <= TOL * max(np.abs(ref).max(), 1e-3 * bound), (kind, ntaps, decim, n, err)


@pytest.mark.parametrize("kind,ntaps,decim", [("ccf", 300, 1), ("ccc", 200, 3), ("ccf", 256, 4), ("ccc", 64, 2), ("ccf", 100, 5),
                                              ("ccc", 400, 20)])
def test_fir_reads_nothing_past_the_guaranteed_items(gpu, po, kind, ntaps, decim):
    """the scheduler guarantees (n-1)*decim + ntaps input items (gr_sync_decimator.cc:46-50); whatever
    follows them (NaNs here) must not reach the outputs -- tiled kernel and overlap-save engine alike"""
    rng = np.random.default_rng(ntaps)
    n = 5000
    need = (n - 1) * decim + ntaps
    x = np.concatenate([_rand_c(rng, need), np.full(decim + 64, np.nan + 1j * np.nan, np.complex64)])
    if kind == "ccf":
        taps = rng.uniform(-1, 1, ntaps).astype(np.float32)
        blk, ref = gpu.fir_filter_ccf(decim, taps), po.fir_ccf(taps, x[:need], n, decim)
    else:
        taps = _rand_c(rng, ntaps)
        blk, ref = gpu.fir_filter_ccc(decim, taps), po.fir_ccc(taps, x[:need], n, decim)
    blk.set_mode(gpu.MODE_FAST)
    got = blk.work(n, x)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= TOL * np.abs(ref).max() * 4


def test_high_decimation_kernel_random_shapes(gpu, po, wl):
    """seeded sweep over the shapes the high-decimation direct kernel takes (every power-of-two factor of the
    decimation selects another LDS layout; output counts around its tiles; two calls per case), for fir_filter_ccf /
    _ccc and freq_xlating with a real (pre-mix form from 16 taps per phase) and a complex prototype"""
    rng = np.random.default_rng(777)
    c = wl.CFG2
    for case in range(40):
        decim = int(rng.choice([3, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32, 48, 64]))
        ntaps = int(rng.choice([1, 2, decim, decim + 1, 3 * decim - 1, 16 * decim, 16 * decim + 3, 25 * decim, 40 * decim]))
        ntaps = min(ntaps, 2000)
        n = int(rng.choice([1, 2, 3, 127, 128, 129, 511, 512, 513, 1000, 2345]))
        kind = ["ccf", "ccc", "xl_real", "xl_cplx"][case % 4]
        nin = n * decim + ntaps - 1
        x = _rand_c(rng, nin)
        n1 = n // 2
        if kind in ("ccf", "ccc"):
            taps = rng.uniform(-1, 1, ntaps).astype(np.float32) if kind == "ccf" else _rand_c(rng, ntaps)
            blk = (gpu.fir_filter_ccf if kind == "ccf" else gpu.fir_filter_ccc)(decim, taps)
            ref = (po.fir_ccf if kind == "ccf" else po.fir_ccc)(taps, x, n, decim)
            bound = np.abs(taps).sum() * np.sqrt(2)
        else:
            taps = wl.lowpass_taps(ntaps, 0.4 / decim, 1.0).astype(np.complex64)
            if kind == "xl_cplx":
                taps = (taps * np.exp(0.02j * np.arange(ntaps))).astype(np.complex64)
            blk = gpu.freq_xlating_fir_filter_ccc(decim, taps, c["center_freq"], c["fs"])
            ref = po.Xlating(decim, taps, c["center_freq"], c["fs"]).work(x, n)
            bound = np.abs(taps).sum() * np.sqrt(2)
        blk.set_mode(gpu.MODE_FAST)
        parts = []
        if n1 > 0:
            parts.append(blk.work(n1, x[: n1 * decim + ntaps - 1]))
        parts.append(blk.work(n - n1, x[n1 * decim:]))
        got = np.concatenate(parts)
        assert got.shape == ref.shape, (kind, ntaps, decim, n)
        err = np.abs(got - ref).max()
        assert err <= TOL * max(np.abs(ref).max(), 1e-3 * bound), (kind, ntaps, decim, n, err)


def test_set_taps_does_not_tear_a_launch_in_flight(gpu, po):
    """the update path rewrites the device tap buffers with blocking null-stream copies; a
    work_device() launch still running on the handle's (non-blocking) stream must be drained first"""
    import torch
    rng = np.random.default_rng(99)
    ntaps, decim, n = 200, 1, 3_000_000
    x = _rand_c(rng, n + ntaps - 1)
    t_old = rng.uniform(-1, 1, ntaps).astype(np.float32)
    t_new = (-t_old[::-1] * 3).astype(np.float32)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d_x = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
    d_y = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    d_y2 = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    blk = gpu.fir_filter_ccf(decim, t_old)
    blk.set_mode(gpu.MODE_GENERIC)           # the slow kernel: several ms in flight
    assert blk.work_device(n, d_x, d_y, st) == n
    blk.set_taps(t_new)
    assert blk.work_device(n, d_x, d_y2, st) == 0        # applies the update, produces nothing (.cc.t:74-79)
    assert blk.work_device(n, d_x, d_y2, st) == n
    st.synchronize()
    got_old = d_y.cpu().numpy().reshape(-1).view(np.complex64)
    got_new = d_y2.cpu().numpy().reshape(-1).view(np.complex64)
    assert bits_equal(got_old, po.fir_ccf(t_old, x, n, decim))
    assert bits_equal(got_new, po.fir_ccf(t_new, x, n, decim))


@pytest.mark.parametrize("mode_name", ["MODE_FAST", "MODE_FAST_VALU"])
def test_fast_demod_parity_numbers_cfg2(gpu, po, wl, mode_name, capsys):
    """the FAST demodulator's error is reported the way it is for the xlating output --
    per element where |ref| is not tiny -- next to the infinity-norm figure, the count of samples that sit on
    the reference's own arctangent step and the transient maximum (parity_util.demod_report)."""
    c = wl.CFG2
    n = 2_000_000
    x = wl.fsk4_capture(n, stream_id=11)
    proto = wl.cfg2_proto_taps()
    nout = n // c["decim"]
    ref = po.chain_xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"], x)
    blk = gpu.xlating_demod(c["decim"], proto, c["center_freq"], c["fs"], c["demod_gain"])
    blk.set_mode(getattr(gpu, mode_name))
    rep = demod_report(blk.work(nout, wl.with_history(x, 255)), ref, gain=c["demod_gain"])
    with capsys.disabled():
        print("\n[%s] cfg2 demod parity: %s" % (mode_name, rep))
    assert rep["ok"]
    assert rep["steady_rel_inf"] <= 1e-5
    # per element over |ref| > 0.1 max|ref|.  Round 3 measured every party against the reference's formula in float64
    # (tools/dbg/demod_attrib.py): the reference's generic build itself sits 8.8e-6 from it and 9.9e-6 from
    # its own SSE build on this capture -- the 1e-5 per-element figure is the reference's reproducibility floor, which only a
    # correctly rounded FIR (or the bit-exact GENERIC mode) can meet against the generic build.  What separates the
    # FAST engines from it is the number of f32 accumulation roundings at full partial-sum magnitude: 256 sequential
    # FMAs in the vector engine (2.42e-5), 30 MFMA accumulations into one tile in round 2 (2.40e-5), ten into each of
    # three tiles that are added at the end since round 3 (1.74e-5).
    assert rep["per_element_rel"] <= (2e-5 if mode_name == "MODE_FAST" else 3e-5)
    assert rep["step_exempted"] <= max(2, int(2e-5 * nout))
    assert rep["transient_rel_inf"] <= 1e-2
