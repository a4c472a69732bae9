"""-m gpu parity tests: clock_recovery_mm_ff (bit-exact), binary_slicer_fb and
correlate_access_code_bb (bit-exact, integer work) through the C ABI."""
import json
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def qa():
    with open(os.path.join(GOLD, "ref_qa_vectors.json")) as f:
        return json.load(f)


def _fsk_soft(rng, nsym, sps=10, noise=0.15):
    x = np.repeat(rng.choice([-3.0, -1.0, 1.0, 3.0], nsym), sps)
    x = np.convolve(x, np.ones(4) / 4, mode="same")
    return (x + rng.normal(0, noise, len(x))).astype(np.float32)


def test_mm_reference_qa(gpu, qa):
    """gr-digital/python/qa_clock_recovery_mm.py:70-102,140-172 on the GPU path"""
    m = qa["mm"]["test02"]
    cr = gpu.clock_recovery_mm_ff(m["omega"], m["gain_omega"], m["mu"], m["gain_mu"], m["omega_rel_lim"])
    out, _ = cr.general_work(1000, np.ones(100, np.float32))
    np.testing.assert_allclose(out[-30:], m["expected_last30"], atol=0.5 * 10 ** -m["places"])
    m = qa["mm"]["test04"]
    cr = gpu.clock_recovery_mm_ff(m["omega"], m["gain_omega"], m["mu"], m["gain_mu"], m["omega_rel_lim"])
    out, _ = cr.general_work(100000, np.array(1000 * [1, 1, -1, -1], np.float32))
    assert np.allclose(np.abs(out[-100:]), 1.31, atol=0.1)


@pytest.mark.parametrize("omega,gm", [(10.0, 0.175), (2.0, 0.05), (4.3, 0.3)])
def test_mm_bit_exact_whole_stream(gpu, po, omega, gm):
    rng = np.random.default_rng(int(omega * 10))
    x = _fsk_soft(rng, 12000, sps=int(round(omega)))
    go = 0.25 * gm * gm
    ref, st = po.chain_mm(omega, go, 0.5, gm, 0.005, x)
    cr = gpu.clock_recovery_mm_ff(omega, go, 0.5, gm, 0.005)
    out, consumed = cr.general_work(len(x), x)
    assert bits_equal(out, ref)
    assert consumed == st["consumed"]
    assert np.float32(cr.mu()) == st["mu"] and np.float32(cr.omega()) == st["omega"]
    # the window refill boundary (4096 floats) is crossed many times here
    assert len(x) > 4 * 4096


def test_mm_chunked_like_scheduler(gpu, po, wl):
    c4 = wl.CFG4
    rng = np.random.default_rng(21)
    x = _fsk_soft(rng, 5000)
    ref, _ = po.chain_mm(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"], x)
    cr = gpu.clock_recovery_mm_ff(c4["omega"], c4["gain_omega"], c4["mu"], c4["gain_mu"], c4["omega_relative_limit"])
    assert cr.forecast(100) == int(np.ceil(100 * c4["omega"] + 8))
    outs, pos = [], 0
    while True:
        o, used = cr.general_work(300, x[pos:pos + 4096])
        if len(o) == 0:
            break
        outs.append(o); pos += used
    assert bits_equal(np.concatenate(outs), ref)


def test_mm_errors_and_setters(gpu, po):
    with pytest.raises(gpu.GrhipError) as e:
        gpu.clock_recovery_mm_ff(0.5, 0.01, 0.5, 0.01, 0.001)
    assert e.value.code == -2          # std::out_of_range
    with pytest.raises(gpu.GrhipError):
        gpu.clock_recovery_mm_ff(2, 0.01, 0.5, -0.01, 0.001)
    cr = gpu.clock_recovery_mm_ff(2, 0.01, 0.5, 0.02, 0.001)
    assert abs(cr.gain_mu() - 0.02) < 1e-9 and abs(cr.gain_omega() - 0.01) < 1e-9
    cr.set_omega(3.0); cr.set_mu(0.25); cr.set_gain_mu(0.05); cr.set_gain_omega(0.002)
    o = po.ClockRecoveryMM(3.0, 0.002, 0.25, 0.05, 0.001)
    x = np.sin(np.arange(3000) * 2 * np.pi / 6.0).astype(np.float32)
    ref, _ = o.general_work(2000, x)
    out, _ = cr.general_work(2000, x)
    assert bits_equal(out, ref)
    # empty / too-short input: nothing produced, nothing consumed
    out, used = cr.general_work(10, np.zeros(5, np.float32))
    assert len(out) == 0 and used == 0


def test_binary_slicer(gpu, po):
    rng = np.random.default_rng(2)
    x = rng.normal(0, 1, 100_001).astype(np.float32)
    x[:4] = [0.0, -0.0, 1e-45, -1e-45]
    assert np.array_equal(gpu.binary_slicer_fb().work(len(x), x), po.binary_slicer_fb(x))


def test_correlator_reference_qa(gpu, qa):
    c = qa["corr"]
    t1 = c["test_001"]
    out = gpu.correlate_access_code_bb(t1["code"], t1["threshold"]).work(len(t1["src"]), np.array(t1["src"], np.uint8))
    assert out.tolist() == t1["expected"]
    code = []
    for b in c["default_access_code_bytes"]:
        code += [(b >> i) & 1 for i in range(8)]
    src = code + c["test_002"]["tail"] + [0] * 64
    exp = [0] * 64 + code + c["test_002"]["expected_tail"]
    out = gpu.correlate_access_code_bb("".join(str(v) for v in code), 0).work(len(src), np.array(src, np.uint8))
    assert out.tolist() == exp
    with pytest.raises(gpu.GrhipError) as e:
        gpu.correlate_access_code_bb("1" * 65, 0)
    assert e.value.code == -2


@pytest.mark.parametrize("L,thr", [(1, 0), (4, 0), (48, 0), (48, 4), (48, 12), (63, 3), (64, 0), (64, 7)])
def test_correlator_random_bit_exact(gpu, po, L, thr):
    rng = np.random.default_rng(L * 100 + thr)
    code = rng.integers(0, 2, L)
    n = 70_001                      # ragged: not a multiple of the 2048-item tile
    bits = rng.integers(0, 256, n).astype(np.uint8)      # only the LSB counts (.cc:124)
    for pos in (0, 5, 2040, 2047, 2048, 4000, 65000, n - L):
        seg = code.copy()
        flip = rng.integers(0, L, min(thr, L))
        seg[flip] ^= 1
        bits[pos:pos + L] = (bits[pos:pos + L] & 0xFE) | seg
    s = "".join(map(str, code))
    ref = po.CorrelateAccessCode(s, thr).work(bits)
    got = gpu.correlate_access_code_bb(s, thr).work(n, bits)
    assert np.array_equal(got, ref)
    assert (ref & 2).any() or L > 60


@pytest.mark.parametrize("shift", [0, 1, 5, 16])
def test_correlator_long_stream_any_alignment(gpu, po, shift):
    """a million items through the device entry: workgroups walk four tiles each with the next tile's loads in flight;
    a stream that starts off a 16-byte boundary takes the ballot path in every tile; two calls on the same handle (the
    state between them now comes from two ballots over the last 128 bits)"""
    import torch
    rng = np.random.default_rng(500 + shift)
    code = rng.integers(0, 2, 48)
    n = 1_000_003
    bits = rng.integers(0, 2, n).astype(np.uint8)
    for pos in rng.integers(0, n - 48, 400):
        bits[pos:pos + 48] = code
    bits[n // 2 - 20:n // 2 + 28] = code          # straddles the split between the two calls
    s = "".join(map(str, code))
    ref = po.CorrelateAccessCode(s, 1).work(bits)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d_in = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    d_in[shift:shift + n] = torch.from_numpy(bits).to(dev)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    blk = gpu.correlate_access_code_bb(s, 1)
    h = n // 2
    assert blk.work_device(h, d_in[shift:], d_out[shift:], st) == h
    assert blk.work_device(n - h, d_in[shift + h:], d_out[shift + h:], st) == n - h
    st.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[shift:shift + n], ref)
    assert not got[:shift].any() and not got[shift + n:].any()


def test_correlator_state_across_calls(gpu, po):
    """d_data_reg / d_flag_reg carried across work() calls, at every alignment"""
    rng = np.random.default_rng(77)
    code = rng.integers(0, 2, 48)
    bits = rng.integers(0, 2, 30_000).astype(np.uint8)
    for pos in range(100, 29_000, 997):
        bits[pos:pos + 48] = code
    s = "".join(map(str, code))
    ref = po.CorrelateAccessCode(s, 2).work(bits)
    blk = gpu.correlate_access_code_bb(s, 2)
    outs, pos = [], 0
    for n in [1, 2, 3, 63, 64, 65, 47, 48, 49, 1000, 2047, 2048, 2049, 4096, 10]:
        outs.append(blk.work(n, bits[pos:pos + n])); pos += n
    outs.append(blk.work(len(bits) - pos, bits[pos:]))
    assert np.array_equal(np.concatenate(outs), ref)


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 100_003])
def test_pager_slicer_bit_exact_and_state(gpu, po, n):
    """pager_slicer_fb: decisions and the DC tracker bit-exact against the oracle, state carried across calls"""
    rng = np.random.default_rng(n)
    # 4-level symbols with a DC offset and noise, so that all four decisions and the tracker matter
    x = (rng.integers(0, 4, n) * 2.0 - 3.0 + 0.7 + 0.3 * rng.standard_normal(n)).astype(np.float32)
    ref_blk = po.PagerSlicer(0.002)
    blk = gpu.pager_slicer_fb(0.002)
    cuts = sorted(set([0, n // 3, (2 * n) // 3, n]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        got = blk.work(b - a, x[a:b])
        assert np.array_equal(got, ref_blk.work(x[a:b]))
    assert blk.dc_offset().tobytes() == ref_blk.dc_offset().tobytes()
    assert len(blk.work(0, np.zeros(0, np.float32))) == 0


@pytest.mark.parametrize("k", [1, 2, 3, 8])
def test_unpack_k_bits(gpu, po, k):
    rng = np.random.default_rng(k)
    x = rng.integers(0, 256, 10_007).astype(np.uint8)
    got = gpu.unpack_k_bits_bb(k).work(len(x) * k, x)
    assert np.array_equal(got, po.unpack_k_bits_bb(k, x))
    # the reference QA vector (gr/qa_unpack_k_bits.py:45-54)
    assert gpu.unpack_k_bits_bb(2).work(8, np.array([2, 3, 0, 1], np.uint8)).tolist() == [1, 0, 1, 1, 0, 0, 0, 1]


def test_unpack_k_bits_errors(gpu):
    with pytest.raises(gpu.GrhipError):
        gpu.unpack_k_bits_bb(0)              # std::out_of_range("interpolation must be > 0")
    with pytest.raises(gpu.GrhipError):
        gpu.unpack_k_bits_bb(2).work(3, np.zeros(2, np.uint8))     # not a multiple of k


def test_four_level_chain_tail(gpu, po, wl):
    """soft 4FSK symbols -> pager_slicer_fb -> unpack_k_bits(2) -> correlate_access_code,
    every stage on the GPU, equal to the oracle chain"""
    rng = np.random.default_rng(77)
    nsym = 50_000
    dibits = rng.integers(0, 4, nsym)
    code = wl.access_code_string()                      # 48 bits = 24 dibits
    code_dibits = [int(code[i:i + 2], 2) for i in range(0, len(code), 2)]
    for start in range(1000, nsym - 100, 5000):
        dibits[start:start + len(code_dibits)] = code_dibits
    soft = (dibits * 2.0 - 3.0 + 0.2 * rng.standard_normal(nsym)).astype(np.float32)
    sl, up = gpu.pager_slicer_fb(0.001), gpu.unpack_k_bits_bb(2)
    corr = gpu.correlate_access_code_bb(code, 0)
    sym = sl.work(nsym, soft)
    bits = up.work(2 * nsym, sym)
    out = corr.work(len(bits), bits)
    o_sym = po.PagerSlicer(0.001).work(soft)
    o_bits = po.unpack_k_bits_bb(2, o_sym)
    o_out = po.CorrelateAccessCode(code, 0).work(o_bits)
    assert np.array_equal(sym, o_sym) and np.array_equal(bits, o_bits) and np.array_equal(out, o_out)
    assert int((out & 2).sum()) // 2 >= 9               # the planted sync words are found


@pytest.mark.parametrize("dtype,nstreams", [(np.complex64, 8), (np.float32, 3), (np.uint8, 5), (np.complex128, 2)])
def test_stream_to_streams_and_back(gpu, dtype, nstreams):
    """gr_stream_to_streams / gr_streams_to_stream (general/gr_stream_to_streams.cc:46-66): a pure
    re-ordering, checked against numpy reshapes; 8-, 4-, 1- and 16-byte items"""
    rng = np.random.default_rng(nstreams)
    n = 10_007
    raw = rng.integers(0, 255, n * nstreams * np.dtype(dtype).itemsize, dtype=np.uint8)
    x = raw.view(dtype).copy()
    s2s = gpu.stream_to_streams(np.dtype(dtype).itemsize, nstreams)
    outs = s2s.work(n, x)
    want = x.view(np.uint8).reshape(n, nstreams, -1)
    for j in range(nstreams):
        assert np.array_equal(outs[j].view(np.uint8).reshape(n, -1), want[:, j, :])
    back = gpu.streams_to_stream(np.dtype(dtype).itemsize, nstreams).work(n * nstreams, outs)
    assert np.array_equal(back.view(np.uint8), x.view(np.uint8))


def test_remaining_harness_adapters(gpu):
    """gr_stream_to_vector (a copy), gr_vector_to_streams (= stream_to_streams), gr_head (WORK_DONE)"""
    import torch
    rng = np.random.default_rng(21)
    x = (rng.standard_normal(4096 * 3) + 1j * rng.standard_normal(4096 * 3)).astype(np.complex64)
    s2v = gpu.stream_to_vector(8, 4096)
    assert np.array_equal(s2v.work(3, x), x)                                   # general/gr_stream_to_vector.cc:46-60
    v2s = gpu.vector_to_streams(8, 4)
    outs = v2s.work(len(x) // 4, x)
    assert all(np.array_equal(outs[j], x[j::4]) for j in range(4))               # general/gr_vector_to_streams.cc:58-65
    h = gpu.head(8, 5000)
    a = h.work(4096, x[:4096])
    b = h.work(4096, x[4096:8192])
    assert len(a) == 4096 and len(b) == 5000 - 4096 and np.array_equal(b, x[4096:5000])
    assert h.work(4096, x[:4096]) is None                                        # general/gr_head.cc:49-50
    h.reset()
    assert len(h.work(10, x[:10])) == 10
    # device pointers: the copy is queued on the caller's stream
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d = torch.from_numpy(x.view(np.float32).reshape(-1, 2)).to(dev)
    o = torch.zeros_like(d)
    h2 = gpu.head(8, 1000)
    assert h2.work_device(4096, d, o, st) == 1000 and h2.work_device(4096, d, o, st) is None
    st.synchronize()
    assert torch.equal(o[:1000], d[:1000]) and float(o[1000:].abs().sum()) == 0.0
    # WORK_DONE has a code of its own (GRHIP_WORK_DONE); an invalid argument is an error, never a quiet end of stream
    h3 = gpu.head(8, 1000)
    with pytest.raises(gpu.GrhipError):
        h3.work_device(16, None, o, st)                  # null input buffer
    with pytest.raises(gpu.GrhipError):
        h3.work_device(-1, d, o, st)
    assert h3.work_device(16, d, o, st) == 16            # the failed calls consumed nothing


def test_binary_slicer_vector_and_tail_paths(gpu, po):
    """16 items per lane on aligned buffers, scalar tail, scalar path on unaligned device pointers; -0.0 slices to 1
    (digital_binary_slicer_fb.cc:54-56: x >= 0)"""
    import torch
    rng = np.random.default_rng(8)
    for n in (1, 15, 16, 17, 4099, 100_003):
        x = rng.standard_normal(n).astype(np.float32)
        x[::7] = 0.0
        x[3::11] = -0.0
        assert np.array_equal(gpu.binary_slicer_fb().work(n, x), po.binary_slicer_fb(x))
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    x = rng.standard_normal(5000).astype(np.float32)
    d = torch.from_numpy(x).to(dev)
    o = torch.zeros(5001, dtype=torch.uint8, device=dev)
    blk = gpu.binary_slicer_fb()
    assert blk.work_device(4999, d[1:], o[1:], st) == 4999           # both pointers off the 16-byte grid
    st.synchronize()
    assert np.array_equal(o[1:5000].cpu().numpy(), po.binary_slicer_fb(x[1:]))


@pytest.mark.parametrize("n", [8191, 8192, 8193, 16383, 16384, 16385, 40_000])
def test_host_entries_across_the_pinned_staging_sizes(gpu, po, n):
    """host-buffer work() calls whose transfers sit on either side of the pinned slots (32 KB) and of the hand-over to the
    runtime's own staging (64 KB): one slot, two slots, runtime path -- same results, call after call on one handle"""
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n).astype(np.float32)              # 4 n bytes in, n bytes out
    bs = gpu.binary_slicer_fb()
    for _ in range(3):
        assert np.array_equal(bs.work(n, x), po.binary_slicer_fb(x))
        x = np.roll(x, 17)
    c = (rng.standard_normal(n + 1) + 1j * rng.standard_normal(n + 1)).astype(np.complex64)    # 8 (n + 1) bytes in, 4 n out
    qd = gpu.quadrature_demod_cf(1.5)
    ref = po.quad_demod_cf(1.5, c, n)
    for _ in range(2):
        assert np.array_equal(qd.work(n, c).view(np.uint32), ref.view(np.uint32))
