"""Known-answer cases for the oracle's gr_framer_sink_1, worked out by hand from
general/gr_framer_sink_1.cc:90-190 and .h:85-98 (the reference ships no QA for this block)."""
import numpy as np


def hdr(length, woff):
    v = ((woff & 0xF) << 12) | (length & 0xFFF)
    return [(v >> (15 - i)) & 1 for i in range(16)] * 2


def test_single_packet_and_flag_is_first_header_bit(po):
    bits = [1, 0, 1] + hdr(2, 9) + [1, 0, 1, 0, 0, 1, 0, 1] + [1, 1, 1, 1, 0, 0, 0, 0] + [1, 1]
    x = np.array(bits, dtype=np.uint8)
    x[3] |= 2                                             # the flagged item carries header bit 31 (.cc:109-112, no count++)
    assert po.FramerSink1().work(x) == [(9, bytes([0xA5, 0xF0]))]


def test_bad_header_zero_length_and_ignored_flags(po):
    h = hdr(1, 3)
    h[20] ^= 1                                            # halves differ -> back to search (.cc:148-149)
    good = hdr(1, 3)
    x = np.array(h + [0] * 5 + hdr(0, 7) + good + [0, 1, 1, 1, 1, 1, 1, 0], dtype=np.uint8)
    x[0] |= 2
    x[37] |= 2                                            # zero-length packet -> empty message with arg1 = 7 (.cc:137-146)
    x[69] |= 2
    x[75] |= 2                                            # inside the header of the third packet: ignored
    x[103] |= 2                                           # inside its payload: ignored
    assert po.FramerSink1().work(x) == [(7, b""), (3, bytes([0x7E]))]


def test_state_carries_across_calls(po):
    x = np.array([0, 0] + hdr(3, 1) + list(np.unpackbits(np.array([1, 2, 3], dtype=np.uint8))) + [0], dtype=np.uint8)
    x[2] |= 2
    f = po.FramerSink1()
    out = []
    for a in range(0, len(x), 5):
        out += f.work(x[a:a + 5])
    assert out == [(1, bytes([1, 2, 3]))]
