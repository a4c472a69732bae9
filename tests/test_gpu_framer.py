"""gr_framer_sink_1 through the C ABI against the oracle's restatement of
general/gr_framer_sink_1.cc:90-190.  The reference holds no QA for this block (grep framer_sink over
its qa_*/test_* files), so the oracle here is pinned only by the hand-computed cases in
tests/test_oracle_framer.py: parity unpinned by reference fixtures.  Bit-exact: same messages, same order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def header_bits(length, woff):
    v = ((woff & 0xF) << 12) | (length & 0xFFF)
    return np.array([(v >> (15 - i)) & 1 for i in range(16)] * 2, dtype=np.uint8)


def make_stream(rng, n, gap=200, maxlen=64, bad=0.15, stray_flags=0.002):
    """correlator-style items: bit 0 data, bit 1 flag on the first header bit"""
    x = rng.integers(0, 2, n, dtype=np.uint8)
    x |= (rng.random(n) < stray_flags).astype(np.uint8) << 1       # flags anywhere: inside payloads too
    pos = int(rng.integers(0, gap))
    while pos + 32 < n:
        ln = int(rng.integers(0, maxlen + 1))
        h = header_bits(ln, int(rng.integers(0, 16)))
        if rng.random() < bad:
            h[int(rng.integers(0, 32))] ^= 1
        x[pos:pos + 32] = (x[pos:pos + 32] & 2) | h
        x[pos] |= 2
        pos += 32 + 8 * ln + int(rng.integers(0, gap))             # gap 0: back to back
    return x


def run_chunks(blk, x, cuts):
    got, a = [], 0
    for b in list(cuts) + [len(x)]:
        if b > a:
            assert blk.work(b - a, x[a:b]) == b - a
        got += blk.messages()
        a = b
    return got


def ref_chunks(po, x, cuts):
    f, out, a = po.FramerSink1(), [], 0
    for b in list(cuts) + [len(x)]:
        out += f.work(x[a:b])
        a = b
    return out


@pytest.mark.parametrize("seed,n,gap,maxlen", [(0, 20000, 200, 64), (1, 100000, 50, 20), (2, 300000, 3000, 700),
                                                  (3, 5000, 1, 3), (4, 70000, 100, 4095), (5, 64, 10, 2), (6, 31, 5, 0)])
def test_framer_matches_oracle_any_chunking(gpu, po, seed, n, gap, maxlen):
    rng = np.random.default_rng(seed)
    x = make_stream(rng, n, gap, maxlen)
    ref = ref_chunks(po, x, [])
    assert ref == ref_chunks(po, x, sorted(rng.integers(0, n, 7)))          # the oracle itself is chunk-invariant
    got = run_chunks(gpu.framer_sink_1(), x, [])
    assert got == ref
    cuts = sorted(int(c) for c in rng.integers(0, n, 9))
    assert run_chunks(gpu.framer_sink_1(), x, cuts) == ref
    # pathological chunking: a few items at a time through headers and payload bytes
    small = list(range(1, min(n, 600), 3))
    assert run_chunks(gpu.framer_sink_1(), x, small) == ref


def test_framer_dense_flags_and_all_zero(gpu, po):
    rng = np.random.default_rng(9)
    x = rng.integers(0, 4, 50000, dtype=np.uint8)                   # a flag on half of the items
    assert run_chunks(gpu.framer_sink_1(), x, [777, 778, 20001]) == ref_chunks(po, x, [])
    z = np.full(4096, 2, dtype=np.uint8)                            # every item flagged, all data bits 0: zero-length packets
    ref = ref_chunks(po, z, [])
    assert len(ref) == 4096 // 32 and all(m == (0, b"") for m in ref)
    assert run_chunks(gpu.framer_sink_1(), z, [100]) == ref


def test_framer_device_path_defers_fetch(gpu, po):
    """work_device() queues several calls; one messages() call returns everything, in order"""
    import torch
    rng = np.random.default_rng(11)
    x = make_stream(rng, 400000, 300, 100)
    ref = ref_chunks(po, x, [])
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    d = torch.from_numpy(x).to(dev)
    blk = gpu.framer_sink_1()
    cuts = [0, 100001, 100002, 250000, 400000]
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert blk.work_device(b - a, d[a:b], st) == b - a
    got = blk.messages(st)
    assert got == ref
    assert blk.messages(st) == []


def test_framer_after_correlator(gpu, po, wl):
    """pkt.py:143-147: correlate_access_code_bb -> framer_sink_1"""
    rng = np.random.default_rng(12)
    code = wl.access_code_string()
    cb = np.array([int(c) for c in code], dtype=np.uint8)
    parts, sent = [], []
    for i in range(40):
        parts.append(rng.integers(0, 2, int(rng.integers(0, 300)), dtype=np.uint8))
        ln = int(rng.integers(0, 50))
        pay = rng.integers(0, 256, ln, dtype=np.uint8)
        woff = int(rng.integers(0, 16))
        parts += [cb, header_bits(ln, woff), np.unpackbits(pay)]
        sent.append((woff, pay.tobytes()))
    parts.append(rng.integers(0, 2, 100, dtype=np.uint8))
    bits = np.concatenate(parts)
    ca = gpu.correlate_access_code_bb(code, 0)
    flagged = ca.work(len(bits), bits)
    ref = po.FramerSink1().work(np.asarray(flagged))
    blk = gpu.framer_sink_1()
    blk.work(len(flagged), flagged)
    got = blk.messages()
    assert got == ref
    # every transmitted packet whose access code was not part of an earlier payload comes out
    assert len(got) >= 35 and all(m in sent for m in got)


def _framer(gpu, seg):
    f = gpu.framer_sink_1()
    f.set_segment_items(seg)
    return f


@pytest.mark.parametrize("seg", [64, 320, 4096, 65536])
@pytest.mark.parametrize("seed,gap,maxlen", [(20, 200, 64), (21, 30, 10), (22, 3000, 900), (23, 1, 4095), (24, 500, 0)])
def test_framer_segment_parallel_walk(gpu, po, seg, seed, gap, maxlen):
    """long calls take the segment-parallel walk (speculative per-segment walks + serial fix-up); set_segment_items
    shrinks the segments so that packets straddle many of them and the fix-up's divergence branch is exercised"""
    rng = np.random.default_rng(seed)
    n = 300_000
    x = make_stream(rng, n, gap, maxlen, stray_flags=0.004)
    ref = ref_chunks(po, x, [])
    assert run_chunks(_framer(gpu, seg), x, []) == ref
    cuts = sorted(int(c) for c in rng.integers(0, n, 4))
    assert run_chunks(_framer(gpu, seg), x, cuts) == ref_chunks(po, x, [])
    dense = rng.integers(0, 4, 100_000, dtype=np.uint8)
    assert run_chunks(_framer(gpu, seg), dense, [33_333]) == ref_chunks(po, dense, [])


def test_framer_parallel_walk_inside_a_long_packet(gpu, po):
    """calls that begin and end inside one 4095-byte packet, on the segment-parallel path (the speculative
    segments of such a call must contribute nothing), then packets again"""
    rng = np.random.default_rng(31)
    x = make_stream(rng, 120_000, 5, 4095, bad=0.0, stray_flags=0.01)
    ref = ref_chunks(po, x, [])
    assert any(len(m[1]) > 2000 for m in ref)
    cuts = list(range(700, len(x), 1000))                     # 1000-item calls: 16 segments each
    assert run_chunks(_framer(gpu, 64), x, cuts) == ref
    cuts = list(range(31, len(x), 257))                       # header bits split across calls as well
    assert run_chunks(_framer(gpu, 64), x, cuts) == ref
