"""-m gpu: the full DMR chain (config 4) as a device-resident multi-stream
pipeline, checked stage by stage against the CPU oracle, plus size-independent
properties at the BASELINE size."""
import numpy as np
import pytest

from conftest import bits_equal, demod_close, rel_err_max

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _oracle_chain(po, wl, x):
    c, c4 = wl.CFG2, wl.CFG4
    dem = po.chain_xlating_demod(c["decim"], wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"], x)
    soft, st = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
    bits = po.binary_slicer_fb(soft)
    out = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(bits)
    return dem, soft, out


def _make_chain(gpu, wl, S, n):
    c, c4 = wl.CFG2, wl.CFG4
    return gpu.dmr_chain(c["decim"], wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"],
                         c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"],
                         wl.access_code_string(), c4["threshold"], S, n)


def test_chain_three_streams_vs_oracle(gpu, po, wl):
    torch = _torch()
    S, n = 3, 600_000
    xs = [wl.fsk4_capture(n, stream_id=40 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + 64
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    nout = n // 4
    d_bits = torch.zeros((S, nout), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    st = torch.cuda.Stream(device=dev)
    import ctypes
    for rep, mode in enumerate((gpu.MODE_FAST, gpu.MODE_FAST, gpu.MODE_GENERIC)):
        # second run: fresh state again, cached rotator table; third: generic order
        ch.set_mode(mode)
        ch.run_device(d_in, n, stride, d_bits, nout, d_n, st)
        st.synchronize()
        nb = d_n.cpu().numpy()
        bits = d_bits.cpu().numpy()
        p_dem, s_dem = ch.intermediate(0)
        p_soft, s_soft = ch.intermediate(1)
        for s in range(S):
            dem_ref, soft_ref, out_ref = _oracle_chain(po, wl, xs[s])
            dem = np.empty(nout, np.float32)
            gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_void_p(p_dem + 4 * s * s_dem), nout * 4)
            assert nb[s] == len(soft_ref)
            soft = np.empty(nb[s], np.float32)
            gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.c_void_p(p_soft + 4 * s * s_soft), int(nb[s]) * 4)
            if mode == gpu.MODE_GENERIC:
                # generic-order FIR => every stage bit-exact, symbols and decisions included
                assert bits_equal(dem, dem_ref)
                assert bits_equal(soft, soft_ref)
            else:
                ok, worst = demod_close(dem, dem_ref)
                assert ok, worst
                # M&M quantises mu to 1/128 sample (rint(mu*128),
                # gri_mmse_fir_interpolator.cc:64): a 1e-6 input difference can
                # pick the neighbouring interpolator phase for one symbol, which
                # moves that symbol by ~slope/128; the loop is contractive so it
                # never diverges.  (The reference's own SSE and generic builds
                # differ from each other in the same way.)
                e = np.abs(soft - soft_ref)
                assert np.median(e) <= 2e-5 and np.quantile(e, 0.99) <= 5e-3 and e.max() <= 0.1
            # bit decisions exact in both modes: no soft symbol is anywhere near the
            # slicer threshold at Es/N0 = 20 dB
            # (the only near-zero symbols are the exact zeros of the filter start-up,
            # which are identical on both sides)
            if mode == gpu.MODE_GENERIC:
                assert np.array_equal(bits[s, :nb[s]], out_ref)
            else:
                # slicer + correlator are exact on whatever symbols they are given ...
                mine = po.CorrelateAccessCode(wl.access_code_string(), wl.CFG4["threshold"]).work(
                    po.binary_slicer_fb(soft))
                assert np.array_equal(bits[s, :nb[s]], mine)
                # ... and the sign decisions can differ from the reference's only where
                # the reference's own soft symbol is within float noise of zero
                flips = po.binary_slicer_fb(soft) != po.binary_slicer_fb(soft_ref)
                assert np.all(np.abs(soft_ref[flips]) < 1e-2) and flips.sum() <= 8
                # every planted sync word is still found (threshold 4 absorbs a stray flip)
                assert abs(int((bits[s, :nb[s]] & 2).sum()) - int((out_ref & 2).sum())) <= 1
                # ... and at the SAME symbol index as in the reference: the flag positions of the
                # two outputs coincide, except a flag the reference itself raises only because of a flipped bit
                mine_pos = np.nonzero(bits[s, :nb[s]] & 2)[0]
                ref_pos = np.nonzero(out_ref & 2)[0]
                planted = ref_pos[np.isin(ref_pos, mine_pos)]
                assert len(planted) >= len(ref_pos) - 1 and len(planted) >= len(mine_pos) - 1
                assert len(ref_pos) >= 5
            assert (out_ref & 2).sum() >= n // 4 // 10 // wl.CFG4["sync_period_syms"]     # sync words found


@pytest.mark.parametrize("decim,ntaps,n_out", [(20, 400, 70_000), (10, 200, 100_000), (5, 200, 150_000), (3, 96, 40_000)])
def test_chain_other_decimations(gpu, po, wl, decim, ntaps, n_out):
    """the chain in front of a decimation the tiled / matrix-core kernels do not take (the direct kernel's batched demodulator
    epilogue; 70 000 and more outputs: in time slices beside the clock recovery, 40 000: one slice): demodulator against the
    oracle, every later stage bit-exact on what the stage before it produced, the planted sync words found in every mode
    (demodulator gain and symbol clock rescaled to the new rate; at 2 samples per symbol, D = 20, the loop does not lock on
    this signal and only the stage-by-stage checks apply)"""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n = 3, n_out * decim + decim - 1
    proto = wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    xs = [wl.fsk4_capture(n, stream_id=50 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + 33
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    omega = c4["omega"] * 4 / decim
    gain = c["demod_gain"] * 4 / decim
    ch = gpu.dmr_chain(decim, proto, c["center_freq"], c["fs"], gain, omega, c4["gain_omega"], c4["mu"],
                       c4["gain_mu"], c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
    st = torch.cuda.Stream(device=dev)
    results = {}
    for mode in (gpu.MODE_FAST, gpu.MODE_GENERIC, gpu.MODE_FAST_VALU):
        ch.set_mode(mode)
        d_bits.zero_()
        ch.run_device(d_in, n, stride, d_bits, n_out, d_n, st)
        st.synchronize()
        nb = d_n.cpu().numpy()
        bits = d_bits.cpu().numpy()
        p_dem, s_dem = ch.intermediate(0)
        p_soft, s_soft = ch.intermediate(1)
        for s in range(S):
            dem_ref = po.chain_xlating_demod(decim, proto, c["center_freq"], c["fs"], gain, xs[s][: n_out * decim])
            dem = np.empty(n_out, np.float32)
            gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * s * s_dem), n_out * 4)
            if mode == gpu.MODE_GENERIC:
                assert bits_equal(dem, dem_ref)
            else:
                ok, worst = demod_close(dem, dem_ref, gain=gain)
                assert ok, (mode, s, worst)
            soft_mine, _ = po.chain_mm(omega, c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
            assert nb[s] == len(soft_mine)
            soft = np.empty(nb[s], np.float32)
            gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft),
                                       int(nb[s]) * 4)
            assert bits_equal(soft, soft_mine)                      # the timing loop is exact on its input, sliced or not
            mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(soft))
            assert np.array_equal(bits[s, :nb[s]], mine)
            results[(mode, s)] = bits[s, :nb[s]].copy()
    if omega >= 4:
        want = n // 40 // c4["sync_period_syms"]
        assert want >= 1
        for key, b in results.items():
            assert want - 1 <= int(np.count_nonzero(b & 2)) <= want + 1, key


@pytest.mark.parametrize("decim,ntaps,n_out,omega", [(4, 256, 175_000, 10.0), (20, 400, 70_000, 2.0), (4, 256, 90_000, 37.3),
                                                     (4, 256, 66_000, 150.7), (4, 256, 40_000, 10.0)])
def test_chain_eight_captures_per_wave(gpu, po, wl, decim, ntaps, n_out, omega):
    """the clock recovery with eight captures per wavefront (the shape of batches beyond a thousand captures, forced here on
    11 captures: one full wave and one with three of its eight groups in use): bit-exact on its input like the one-capture
    form, and the two forms produce identical chains.  Symbol clocks from 2 to 150 samples per symbol: many symbols per
    ring top-up, few, and (150 > the ring's chunk) a ring re-seeded at every symbol; time-sliced and single-slice runs."""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n = 11, n_out * decim + 1
    proto = wl.cfg2_proto_taps() if decim == 4 else wl.lowpass_taps(ntaps, 100e3, 10e6).astype(np.complex64)
    gain = c["demod_gain"] * 4 / decim
    xs = [wl.fsk4_capture(n, stream_id=60 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + 7
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = gpu.dmr_chain(decim, proto, c["center_freq"], c["fs"], gain, omega, c4["gain_omega"], c4["mu"], c4["gain_mu"],
                       c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
    st = torch.cuda.Stream(device=dev)
    for mode in (gpu.MODE_FAST, gpu.MODE_GENERIC):
        ch.set_mode(mode)
        got = {}
        for cpw in (8, 32, 1):
            ch.set_captures_per_wave(cpw)
            d_bits.zero_(); d_n.zero_()
            torch.cuda.synchronize()
            ch.run_device(d_in, n, stride, d_bits, n_out, d_n, st)
            st.synchronize()
            nb = d_n.cpu().numpy()
            bits = d_bits.cpu().numpy()
            p_dem, s_dem = ch.intermediate(0)
            p_soft, s_soft = ch.intermediate(1)
            softs = []
            for s in range(S):
                dem = np.empty(n_out, np.float32)
                gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * s * s_dem), n_out * 4)
                soft = np.empty(nb[s], np.float32)
                gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft),
                                           int(nb[s]) * 4)
                if cpw != 1:
                    ref, _ = po.chain_mm(omega, c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
                    assert nb[s] == len(ref), (mode, s)
                    assert bits_equal(soft, ref), (mode, s)
                    mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(soft))
                    assert np.array_equal(bits[s, :nb[s]], mine)
                softs.append((soft, bits[s, :nb[s]].copy()))
            got[cpw] = softs
        for s in range(S):
            assert bits_equal(got[8][s][0], got[1][s][0]) and np.array_equal(got[8][s][1], got[1][s][1]), (mode, s)
            assert bits_equal(got[32][s][0], got[1][s][0]) and np.array_equal(got[32][s][1], got[1][s][1]), (mode, s)
    # the 4FSK tail behind it (reads the symbol counts the clock recovery leaves)
    ch.set_mode(gpu.MODE_FAST)
    ch.set_four_level(True, 0.01)
    d_bits2 = torch.zeros((S, 2 * n_out), dtype=torch.uint8, device=dev)
    outs = {}
    for cpw in (8, 32, 1):
        ch.set_captures_per_wave(cpw)
        d_bits2.zero_()
        torch.cuda.synchronize()
        ch.run_device(d_in, n, stride, d_bits2, 2 * n_out, d_n, st)
        st.synchronize()
        outs[cpw] = (d_n.cpu().numpy().copy(), d_bits2.cpu().numpy().copy())
    assert np.array_equal(outs[8][0], outs[1][0]) and np.array_equal(outs[8][1], outs[1][1])
    assert np.array_equal(outs[32][0], outs[1][0]) and np.array_equal(outs[32][1], outs[1][1])


@pytest.mark.parametrize("n_out,omega", [(70_000, 10.0), (66_000, 37.3), (40_000, 3.3)])
def test_chain_thirty_two_captures_per_wave(gpu, po, wl, n_out, omega):
    """the clock recovery with 32 captures per wavefront, two lanes each, its samples through a FIFO in registers
    (mm_pairs_kernel): 37 captures = one full wave and one with five pairs in use.  The captures of a wave take their
    chunks in the same pass, so captures whose symbol clocks differ drift apart in the ring until the wave starts its FIFO
    again: symbol rates of 39 / 40 / 41 samples per symbol (a loop that sits on its omega limit either side), one capture
    of noise only, one that ends early (zeros).  Bit-exact on its input against the oracle, equal to the one-capture
    form; time-sliced (FAST) and single-slice (GENERIC) runs."""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n = 37, n_out * 4 + 1
    proto = wl.cfg2_proto_taps()
    gain = c["demod_gain"]
    xs = []
    for s in range(S):
        cfg = dict(c)
        cfg["sym_rate"] = c["fs"] / (39 + s % 3)
        x = wl.fsk4_capture(n, stream_id=200 + s, cfg=cfg)
        if s == 7:
            rng = np.random.default_rng(7)
            x = (rng.normal(0, 0.5, n) + 1j * rng.normal(0, 0.5, n)).astype(np.complex64)
        if s == 9:
            x[n // 3:] = 0
        xs.append(x)
    dev = torch.device("cuda", 0)
    stride = n + 7
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = gpu.dmr_chain(4, proto, c["center_freq"], c["fs"], gain, omega, c4["gain_omega"], c4["mu"], c4["gain_mu"],
                       c4["omega_relative_limit"], wl.access_code_string(), c4["threshold"], S, n)
    st = torch.cuda.Stream(device=dev)
    for mode in (gpu.MODE_FAST, gpu.MODE_GENERIC):
        ch.set_mode(mode)
        got = {}
        for cpw in (32, 1):
            ch.set_captures_per_wave(cpw)
            d_bits.zero_(); d_n.zero_()
            torch.cuda.synchronize()
            ch.run_device(d_in, n, stride, d_bits, n_out, d_n, st)
            st.synchronize()
            nb = d_n.cpu().numpy()
            bits = d_bits.cpu().numpy()
            p_dem, s_dem = ch.intermediate(0)
            p_soft, s_soft = ch.intermediate(1)
            softs = []
            for s in range(S):
                soft = np.empty(nb[s], np.float32)
                gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft),
                                           int(nb[s]) * 4)
                if cpw == 32 and s in (0, 1, 2, 7, 9, 31, 32, 36):
                    dem = np.empty(n_out, np.float32)
                    gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * s * s_dem), n_out * 4)
                    ref, _ = po.chain_mm(omega, c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
                    assert nb[s] == len(ref), (mode, s)
                    assert bits_equal(soft, ref), (mode, s)
                softs.append((soft, bits[s, :nb[s]].copy()))
            got[cpw] = softs
        for s in range(S):
            assert len(got[32][s][0]) == len(got[1][s][0]), (mode, s)
            assert bits_equal(got[32][s][0], got[1][s][0]) and np.array_equal(got[32][s][1], got[1][s][1]), (mode, s)


@pytest.mark.parametrize("limit", [1, 7, 8, 9, 1003, 4096, 6900])
def test_clock_recovery_stops_at_its_output_limit(gpu, po, wl, limit):
    """`oo < noutput_items` (digital_clock_recovery_mm_ff.cc:113) with a limit that is not a multiple of the
    eight symbols a pass of mm_rows_kernel takes: both forms of the loop stop at exactly `limit` symbols, with the
    oracle's symbols (the eight-captures form used to run whole passes of eight past the limit)"""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n_out = 11, 70_000
    n = n_out * 4
    xs = [wl.fsk4_capture(n, stream_id=40 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    d_in = torch.zeros((S, n, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    ch.set_mode(gpu.MODE_GENERIC)
    ch.set_max_symbols(limit)
    st = torch.cuda.Stream(device=dev)
    got = {}
    for cpw in (8, 32, 1):
        ch.set_captures_per_wave(cpw)
        d_bits.zero_(); d_n.zero_()
        torch.cuda.synchronize()
        ch.run_device(d_in, n, n, d_bits, n_out, d_n, st)
        st.synchronize()
        nb = d_n.cpu().numpy()
        assert (nb == limit).all(), (cpw, nb)
        p_soft, s_soft = ch.intermediate(1)
        softs = []
        for s in range(S):
            soft = np.empty(limit, np.float32)
            gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft), limit * 4)
            softs.append(soft)
        got[cpw] = (softs, d_bits.cpu().numpy()[:, :limit].copy())
    dem_ref, _, _ = _oracle_chain(po, wl, xs[3])
    ref, _ = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem_ref)
    assert bits_equal(got[8][0][3], ref[:limit])
    for s in range(S):
        assert bits_equal(got[8][0][s], got[1][0][s]), s
        assert bits_equal(got[32][0][s], got[1][0][s]), s
    assert np.array_equal(got[8][1], got[1][1])
    assert np.array_equal(got[32][1], got[1][1])


@pytest.mark.parametrize("S", [1600, 2100])
def test_chain_big_batches_take_the_eight_captures_form_by_themselves(gpu, po, wl, S):
    """batches of 1600 captures and more: the library's own choice of the clock-recovery form (eight captures per wave, the
    loop and the FIR on CUs of their own; 2100 captures: the smaller ring), time-sliced.  Few distinct captures repeated over
    the batch: every copy must give what its original gives, the originals are checked stage by stage, and forcing one wave
    per capture reproduces the batch bit for bit."""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    n_out = 70_000
    n = n_out * 4
    K = 5                                               # distinct captures
    xs = [wl.fsk4_capture(n, stream_id=80 + k) for k in range(K)]
    dev = torch.device("cuda", 0)
    d_in = torch.empty((S, n, 2), dtype=torch.float32, device=dev)
    src = [torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev) for x in xs]
    for s in range(S):
        d_in[s] = src[s % K]
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    ch.run_device(d_in, n, n, d_bits, n_out, d_n, st)
    st.synchronize()
    nb = d_n.cpu().numpy()
    bits = d_bits.cpu().numpy()
    p_dem, s_dem = ch.intermediate(0)
    p_soft, s_soft = ch.intermediate(1)
    for k in range(K):
        dem = np.empty(n_out, np.float32)
        gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * k * s_dem), n_out * 4)
        dem_ref, _, _ = _oracle_chain(po, wl, xs[k])
        ok, worst = demod_close(dem, dem_ref)
        assert ok, (k, worst)
        ref, _ = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
        assert nb[k] == len(ref)
        soft = np.empty(nb[k], np.float32)
        gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * k * s_soft), int(nb[k]) * 4)
        assert bits_equal(soft, ref), k
        mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(soft))
        assert np.array_equal(bits[k, :nb[k]], mine)
    for s in range(K, S):                               # every copy like its original
        assert nb[s] == nb[s % K] and np.array_equal(bits[s, :nb[s]], bits[s % K, :nb[s]]), s
    first = bits.copy()
    ch.set_captures_per_wave(1)
    d_bits.zero_()
    torch.cuda.synchronize()
    ch.run_device(d_in, n, n, d_bits, n_out, d_n, st)
    st.synchronize()
    assert np.array_equal(d_n.cpu().numpy(), nb) and np.array_equal(d_bits.cpu().numpy(), first)


def test_chain_full_size_captures_in_a_big_batch(gpu, po, wl):
    """the benchmarked shape at its size -- 1600 captures of 10 M samples (a few distinct ones, repeated
    with different neighbours in their wavefronts) through the library's own choice of the clock-recovery form
    (mm_rows_kernel<1024>, the loop and the FIR on CUs of their own, 32 time slices, ~4900 ring top-ups per capture).
    The originals are checked stage by stage against the oracle: demodulator within the FAST tolerance, symbols exact on
    that input, bit decisions exact on those symbols; every copy equals its original bit for bit."""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n = 1600, 10_000_000
    n_out = n // 4
    K = 3
    free_b, _ = torch.cuda.mem_get_info(0)
    if free_b < S * (n * 8 + n_out * 10) + (8 << 30):
        pytest.skip("needs ~150 GB of device memory")
    xs = [wl.fsk4_capture(n, stream_id=300 + k) for k in range(K)]
    dev = torch.device("cuda", 0)
    d_in = torch.empty((S, n, 2), dtype=torch.float32, device=dev)
    src = [torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev) for x in xs]
    order = [(s * 7 + s // 8) % K for s in range(S)]        # neighbours in a wave of eight differ from wave to wave
    for s in range(S):
        d_in[s] = src[order[s]]
    del src
    d_bits = torch.zeros((S, n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    ch.run_device(d_in, n, n, d_bits, n_out, d_n, st)
    st.synchronize()
    nb = d_n.cpu().numpy()
    p_dem, s_dem = ch.intermediate(0)
    p_soft, s_soft = ch.intermediate(1)
    first = {}
    for s in range(S):
        first.setdefault(order[s], s)
    for k in range(K):
        s0 = first[k]
        dem = np.empty(n_out, np.float32)
        gpu.lib().grhip_memcpy_d2h(dem.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_dem + 4 * s0 * s_dem), n_out * 4)
        dem_ref = po.chain_xlating_demod(c["decim"], wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"], xs[k])
        ok, worst = demod_close(dem, dem_ref)
        assert ok, (k, worst)
        ref, _ = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], dem)
        assert nb[s0] == len(ref), k
        soft = np.empty(nb[s0], np.float32)
        gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s0 * s_soft), int(nb[s0]) * 4)
        assert bits_equal(soft, ref), k
        mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(soft))
        b0 = d_bits[s0, :int(nb[s0])].cpu().numpy()
        assert np.array_equal(b0, mine), k
        want = len(range(100, len(ref) - 48, c4["sync_period_syms"]))
        assert want - 1 <= int(np.count_nonzero(b0 & 2)) <= want + 1
    # every copy like its original (compared on the device: 4 GB of decisions)
    for k in range(K):
        rows = torch.tensor([s for s in range(S) if order[s] == k], device=dev)
        assert bool((d_n[rows] == int(nb[first[k]])).all())
        assert bool((d_bits[rows] == d_bits[first[k]]).all()), k


def test_chain_more_symbols_than_the_nominal_clock_gives(gpu, po, wl):
    """the stages behind the clock recovery size their grids by the nominal symbol count (+ 12.5 %) and walk longer streams
    in strides: a 2-samples-per-symbol loop allowed to go down to 1.1 (the limit is absolute: digital_clock_recovery_mm_ff.cc:124),
    on a frequency waveform that makes every timing error negative (pulses decaying to zero with alternating sign), settles
    there and produces 1.2 times the nominal count -- every symbol is still sliced and correlated (both clock-recovery
    forms, binary and 4-level tails)"""
    import ctypes
    torch = _torch()
    c, c4 = wl.CFG2, wl.CFG4
    S, n_out = 3, 800_000
    n = n_out * 4
    xs = []
    for s in range(S):
        P = 400 + 40 * s
        t = np.arange(n)
        saw = (1.0 - (t % P) / P) * np.where((t // P) % 2 == 0, 1.0, -1.0)
        ph = 2 * np.pi * np.cumsum(c["center_freq"] + 30e3 * saw) / c["fs"]
        xs.append(np.exp(1j * ph).astype(np.complex64))
    dev = torch.device("cuda", 0)
    d_in = torch.zeros((S, n, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    d_bits = torch.zeros((S, 2 * n_out), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    omega, g_omega, g_mu, rel = 2.0, 0.1, 0.01, 0.9
    ch = gpu.dmr_chain(4, wl.cfg2_proto_taps(), c["center_freq"], c["fs"], c["demod_gain"], omega, g_omega, c4["mu"], g_mu, rel,
                       wl.access_code_string(), c4["threshold"], S, n)
    st = torch.cuda.Stream(device=dev)
    longest = 0
    for cpw in (1, 8):
        ch.set_captures_per_wave(cpw)
        ch.set_four_level(False, 0.0)
        d_bits.zero_()
        torch.cuda.synchronize()
        ch.run_device(d_in, n, n, d_bits, 2 * n_out, d_n, st)
        st.synchronize()
        nb = d_n.cpu().numpy()
        bits = d_bits.cpu().numpy()
        p_soft, s_soft = ch.intermediate(1)
        softs = []
        for s in range(S):
            soft = np.empty(nb[s], np.float32)
            gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft), int(nb[s]) * 4)
            mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(po.binary_slicer_fb(soft))
            assert np.array_equal(bits[s, :nb[s]], mine), (cpw, s)
            softs.append(soft)
            longest = max(longest, int(nb[s]))
        ch.set_four_level(True, 0.01)
        d_bits.zero_()
        torch.cuda.synchronize()
        ch.run_device(d_in, n, n, d_bits, 2 * n_out, d_n, st)
        st.synchronize()
        nb2 = d_n.cpu().numpy()
        bits2 = d_bits.cpu().numpy()
        for s in range(S):
            assert nb2[s] == 2 * nb[s]
            dib = po.unpack_k_bits_bb(2, po.PagerSlicer(0.01).work(softs[s]))
            mine = po.CorrelateAccessCode(wl.access_code_string(), c4["threshold"]).work(dib)
            assert np.array_equal(bits2[s, :nb2[s]], mine), (cpw, s)
    # the strided walk was taken: more symbols than the grid covers (workgroups of 4 tiles of 8192 items)
    assert longest > -(-int(n_out / omega * 1.125 + 4096) // 32768) * 32768, longest


def test_chain_properties_full_size(gpu, wl):
    """one 10 M-sample capture (BASELINE size): sync flags appear every
    sync_period symbols, streams are independent and runs are reproducible."""
    torch = _torch()
    n = 10_000_000
    x = wl.fsk4_capture(n, stream_id=0)
    dev = torch.device("cuda", 0)
    S = 2
    d_in = torch.zeros((S, n, 2), dtype=torch.float32, device=dev)
    d_in[0] = torch.from_numpy(x.view(np.float32).reshape(-1, 2))
    d_in[1] = d_in[0]
    nout = n // 4
    d_bits = torch.zeros((S, nout), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    st = torch.cuda.Stream(device=dev)
    ch.run_device(d_in, n, n, d_bits, nout, d_n, st)
    st.synchronize()
    nb = d_n.cpu().numpy()
    assert nb[0] == nb[1] and abs(nb[0] - n / 40) < 200
    b = d_bits.cpu().numpy()
    assert np.array_equal(b[0, :nb[0]], b[1, :nb[1]])          # identical inputs, identical streams
    hits = np.nonzero(b[0, :nb[0]] & 2)[0]
    per = wl.CFG4["sync_period_syms"]
    assert len(hits) >= n // 40 // per - 1
    d = np.diff(hits)
    assert np.all(np.abs(d[d > 100] - per) <= 2)               # flags one sync period apart
    first = b[0, :nb[0]].copy()
    ch.run_device(d_in, n, n, d_bits, nout, d_n, st)
    st.synchronize()
    assert np.array_equal(d_bits.cpu().numpy()[0, :nb[0]], first)   # idempotent


def test_four_level_tail_in_the_batched_chain(gpu, po, wl):
    """pager_slicer_fb -> unpack_k_bits(2) -> correlate_access_code inside the multi-capture chain
    (S > 1, per-stream symbol counts), then the multi-capture framer_sink_1 on its output.  GENERIC mode: every
    stage bit-exact against the oracle chain; FAST mode: the tail is exact given the symbols the chain produced."""
    import ctypes
    torch = _torch()
    S, n = 3, 400_000
    alpha = 0.002
    xs = [wl.fsk4_capture(n, stream_id=70 + s) for s in range(S)]
    dev = torch.device("cuda", 0)
    stride = n + 64
    d_in = torch.zeros((S, stride, 2), dtype=torch.float32, device=dev)
    for s in range(S):
        d_in[s, :n] = torch.from_numpy(xs[s].view(np.float32).reshape(-1, 2))
    nout = n // 4
    d_bits = torch.zeros((S, 2 * nout), dtype=torch.uint8, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    ch = _make_chain(gpu, wl, S, n)
    ch.set_four_level(True, alpha)
    fr = gpu.framer_sink_1_batch(S, 2 * nout)
    st = torch.cuda.Stream(device=dev)
    code, thr = wl.access_code_string(), wl.CFG4["threshold"]
    for mode in (gpu.MODE_FAST, gpu.MODE_GENERIC):
        ch.set_mode(mode)
        ch.run_device(d_in, n, stride, d_bits, 2 * nout, d_n, st)
        fr.run_device(d_bits, 2 * nout, d_n, 2 * nout, st)
        msgs = fr.messages(st)
        nb = d_n.cpu().numpy()
        bits = d_bits.cpu().numpy()
        p_soft, s_soft = ch.intermediate(1)
        p_sym, s_sym = ch.intermediate(2)
        for s in range(S):
            dem_ref, soft_ref, _ = _oracle_chain(po, wl, xs[s])
            nsym = int(nb[s]) // 2
            assert nb[s] == 2 * nsym and nsym == len(soft_ref)
            soft = np.empty(nsym, np.float32)
            gpu.lib().grhip_memcpy_d2h(soft.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_soft + 4 * s * s_soft), nsym * 4)
            sym = np.empty(nsym, np.uint8)
            gpu.lib().grhip_memcpy_d2h(sym.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(p_sym + s * s_sym), nsym)
            src = soft_ref if mode == gpu.MODE_GENERIC else soft          # FAST: exact given its own symbols
            if mode == gpu.MODE_GENERIC:
                assert bits_equal(soft, soft_ref)
            o_sym = po.PagerSlicer(alpha).work(src)
            o_bits = po.unpack_k_bits_bb(2, o_sym)
            o_out = po.CorrelateAccessCode(code, thr).work(o_bits)
            assert np.array_equal(sym, o_sym)
            assert np.array_equal(bits[s, :nb[s]], o_out)
            assert msgs[s] == po.FramerSink1().work(o_out)
            if mode == gpu.MODE_FAST:
                # against the reference's own symbols: decisions differ only where a soft symbol sits on a threshold
                r_sym = po.PagerSlicer(alpha).work(soft_ref)
                assert (r_sym != sym).mean() <= 1e-4


def test_framer_batch_matches_oracle_per_stream(gpu, po):
    """multi-capture framer_sink_1: streams of different lengths (device-side counts), packets of every size,
    bad headers, stray flags; each stream framed from the search state, equal to the oracle stream by stream"""
    torch = _torch()
    from test_gpu_framer import make_stream
    rng = np.random.default_rng(5)
    lens = [150_001, 31, 0, 70_000, 4096, 99_999, 1]
    S, stride = len(lens), 150_016
    xs = [make_stream(rng, L, gap=(50, 10, 1, 3000, 100, 1, 1)[i], maxlen=(90, 0, 0, 700, 4095, 3, 0)[i]) for i, L in enumerate(lens)]
    dev = torch.device("cuda", 0)
    d_in = torch.zeros((S, stride), dtype=torch.uint8, device=dev)
    for s in range(S):
        d_in[s, :lens[s]] = torch.from_numpy(xs[s])
    d_n = torch.tensor(lens, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    fr = gpu.framer_sink_1_batch(S, stride)
    for rep in range(2):            # every run starts from the search state
        fr.run_device(d_in, stride, d_n, stride, st)
        got = fr.messages(st)
        for s in range(S):
            assert got[s] == po.FramerSink1().work(xs[s]), s
    assert sum(len(m) for m in got) > 1000
    # without a count array every stream has n_items_max items
    fr.run_device(d_in, stride, None, 4096, st)
    got = fr.messages(st)
    for s in range(S):
        assert got[s] == po.FramerSink1().work(np.concatenate([xs[s], np.zeros(stride, np.uint8)])[:4096]), s
