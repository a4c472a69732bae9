"""ctypes front-end for the TEST-ONLY checkers (oracle/liboracle.so and, when
present, oracle/_ref/libgrref.so).  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def _c64_as_f32(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a.view(np.float32)


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_orc = _load(os.path.join(_HERE, "liboracle.so"))
_ref = _load(os.path.join(_HERE, "_ref", "libgrref.so"))


def have_oracle():
    return _orc is not None


def have_ref():
    return _ref is not None


def _need():
    if _orc is None:
        raise RuntimeError("oracle/liboracle.so not built: run `make -C oracle`")
    return _orc


# ----------------------------------------------------------------------------
# oracle (this repo's restatement)
# ----------------------------------------------------------------------------
def _fir(fn, taps, tap_c, x, x_c, n, decim):
    o = _need()
    f = getattr(o, fn)
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]
    taps = np.ascontiguousarray(taps, dtype=np.complex64 if tap_c else np.float32)
    x = np.ascontiguousarray(x, dtype=np.complex64 if x_c else np.float32)
    ntaps = len(taps)
    need = (n - 1) * decim + ntaps if n > 0 else 0
    assert len(x) >= need, (len(x), need)
    out = np.zeros(n, dtype=np.complex64 if x_c else np.float32)
    f(taps.ctypes.data, ntaps, x.ctypes.data, out.ctypes.data, n, decim)
    return out


def fir_fff(taps, x, n, decim=1):
    return _fir("orc_fir_fff", taps, False, x, False, n, decim)


def fir_ccf(taps, x, n, decim=1):
    return _fir("orc_fir_ccf", taps, False, x, True, n, decim)


def fir_ccc(taps, x, n, decim=1):
    return _fir("orc_fir_ccc", taps, True, x, True, n, decim)


class FirFilterWithBuffer(object):
    """gri_fir_filter_with_buffer_{fff,ccf,ccc} (filter/gri_fir_filter_with_buffer_XXX.cc.t:30-121): the filter keeps
    its own delay line across calls; filterNdec(x, n, dec) consumes n * dec items"""
    KINDS = {"fff": 0, "ccf": 1, "ccc": 2}

    def __init__(self, kind, taps):
        o = _need()
        self.kind = kind
        self._k = self.KINDS[kind]
        self._tap_dt = np.complex64 if kind == "ccc" else np.float32
        self._io_dt = np.float32 if kind == "fff" else np.complex64
        o.orc_fwb_create.restype = C.c_void_p
        o.orc_fwb_create.argtypes = [C.c_int, C.c_void_p, C.c_uint]
        o.orc_fwb_destroy.argtypes = [C.c_void_p]
        o.orc_fwb_filterNdec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]
        o.orc_fwb_filterNdec.restype = None
        self._o = o
        self._h = None
        self.set_taps(taps)

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=self._tap_dt)
        if self._h:
            self._o.orc_fwb_destroy(self._h)
        self._h = self._o.orc_fwb_create(self._k, t.ctypes.data, len(t))

    def filterNdec(self, x, n, dec=1):
        x = np.ascontiguousarray(x, dtype=self._io_dt)
        assert len(x) >= n * dec
        out = np.zeros(n, dtype=self._io_dt)
        self._o.orc_fwb_filterNdec(self._h, x.ctypes.data, out.ctypes.data, n, dec)
        return out

    def __del__(self):
        try:
            if self._h:
                self._o.orc_fwb_destroy(self._h)
        except Exception:
            pass


def fast_atan2f(y, x):
    o = _need()
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    out = np.empty_like(y)
    o.orc_fast_atan2f_n.restype = None
    o.orc_fast_atan2f_n.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
    o.orc_fast_atan2f_n(y, x, out, len(y))
    return out


class _Rot(C.Structure):
    _fields_ = [("pr", C.c_float), ("pi", C.c_float), ("ir", C.c_float), ("ii", C.c_float),
                ("counter", C.c_uint)]


def rotator_phases(incr, n):
    """phases used for outputs 0..n-1 by a fresh gr_rotator after set_phase_incr(incr)."""
    o = _need()
    r = _Rot()
    o.orc_rotator_init(C.byref(r))
    o.orc_rotator_set_phase_incr.argtypes = [C.c_void_p, C.c_float, C.c_float]
    o.orc_rotator_set_phase_incr(C.byref(r), np.float32(incr.real), np.float32(incr.imag))
    out = np.zeros(n, dtype=np.complex64)
    o.orc_rotator_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    o.orc_rotator_phases(C.byref(r), out.ctypes.data, n)
    return out


class Xlating:
    """gr_freq_xlating_fir_filter_ccc restatement (stateful rotator)."""

    def __init__(self, decim, proto, center_freq, sampling_freq):
        o = _need()
        self.proto = np.ascontiguousarray(proto, dtype=np.complex64)
        self.decim = int(decim)
        o.orc_xlating_ccc_new.restype = C.c_void_p
        o.orc_xlating_ccc_new.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_double, C.c_double]
        self.h = o.orc_xlating_ccc_new(self.decim, self.proto.ctypes.data, len(self.proto),
                                       float(center_freq), float(sampling_freq))
        self.ntaps = len(self.proto)

    def ctaps(self):
        o = _need()
        o.orc_xlating_ctaps.restype = C.POINTER(C.c_float)
        o.orc_xlating_ctaps.argtypes = [C.c_void_p]
        p = o.orc_xlating_ctaps(self.h)
        return np.ctypeslib.as_array(p, shape=(2 * self.ntaps,)).copy().view(np.complex64)

    def rot(self):
        o = _need()
        five = np.zeros(5, dtype=np.float32)
        o.orc_xlating_get_rot.argtypes = [C.c_void_p, _f32p]
        o.orc_xlating_get_rot(self.h, five)
        return complex(five[0], five[1]), complex(five[2], five[3]), int(five[4])

    def work(self, x_with_history, nout):
        """x_with_history holds (nout-1)*decim + ntaps items."""
        o = _need()
        x = np.ascontiguousarray(x_with_history, dtype=np.complex64)
        need = (nout - 1) * self.decim + self.ntaps if nout else 0
        assert len(x) >= need
        out = np.zeros(nout, dtype=np.complex64)
        o.orc_xlating_ccc_work.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        o.orc_xlating_ccc_work(self.h, x.ctypes.data, out.ctypes.data, nout)
        return out

    def __del__(self):
        try:
            _orc.orc_xlating_free.argtypes = [C.c_void_p]
            _orc.orc_xlating_free(self.h)
        except Exception:
            pass


def quad_demod_cf(gain, x_with_history, nout):
    """x_with_history: nout+1 complex items (first is the history item)."""
    o = _need()
    x = np.ascontiguousarray(x_with_history, dtype=np.complex64)
    assert len(x) >= nout + 1
    out = np.zeros(nout, dtype=np.float32)
    o.orc_quad_demod_cf.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t]
    o.orc_quad_demod_cf(np.float32(gain), x.ctypes.data, out.ctypes.data, nout)
    return out


def mmse_taps_reversed():
    o = _need()
    o.orc_mmse_taps_reversed.restype = C.POINTER(C.c_float)
    return np.ctypeslib.as_array(o.orc_mmse_taps_reversed(), shape=(129, 8)).copy()


def mmse_interpolate(x8, mu):
    o = _need()
    x8 = np.ascontiguousarray(x8, dtype=np.float32)
    o.orc_mmse_interpolate.restype = C.c_float
    o.orc_mmse_interpolate.argtypes = [_f32p, C.c_float]
    return np.float32(o.orc_mmse_interpolate(x8, np.float32(mu)))


def branchless_clip(x, clip):
    o = _need()
    o.orc_branchless_clip.restype = C.c_float
    o.orc_branchless_clip.argtypes = [C.c_float, C.c_float]
    return np.array([o.orc_branchless_clip(np.float32(v), np.float32(clip)) for v in np.ravel(x)],
                    dtype=np.float32)


class _MM(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("mu", "omega", "min_omega", "omega_mid", "max_omega",
                                         "gain_omega", "gain_mu", "last_sample",
                                         "omega_relative_limit")]


class ClockRecoveryMM:
    """digital_clock_recovery_mm_ff restatement."""

    def __init__(self, omega, gain_omega, mu, gain_mu, omega_relative_limit):
        o = _need()
        self.s = _MM()
        o.orc_mm_init.argtypes = [C.c_void_p] + [C.c_float] * 5
        rc = o.orc_mm_init(C.byref(self.s), omega, gain_omega, mu, gain_mu, omega_relative_limit)
        if rc:
            raise IndexError("out_of_range")  # std::out_of_range in the reference

    def forecast(self, nout):
        o = _need()
        o.orc_mm_forecast.argtypes = [C.c_void_p, C.c_int]
        return o.orc_mm_forecast(C.byref(self.s), nout)

    def general_work(self, nout, x):
        """returns (out[:n], consumed)"""
        o = _need()
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(max(nout, 1), dtype=np.float32)
        consumed = C.c_int(0)
        o.orc_mm_general_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int)]
        n = o.orc_mm_general_work(C.byref(self.s), nout, len(x), x.ctypes.data, out.ctypes.data,
                                  C.byref(consumed))
        return out[:n].copy(), consumed.value

    @property
    def state(self):
        return dict(mu=np.float32(self.s.mu), omega=np.float32(self.s.omega),
                    last_sample=np.float32(self.s.last_sample),
                    omega_mid=np.float32(self.s.omega_mid))


class ClockRecoveryMMcc:
    """digital_clock_recovery_mm_cc restatement."""

    def __init__(self, omega, gain_omega, mu, gain_mu, omega_relative_limit):
        o = _need()
        o.orc_mmcc_size.restype = C.c_size_t
        self.s = C.create_string_buffer(o.orc_mmcc_size())
        o.orc_mmcc_init.argtypes = [C.c_void_p] + [C.c_float] * 5
        if o.orc_mmcc_init(self.s, omega, gain_omega, mu, gain_mu, omega_relative_limit):
            raise IndexError("out_of_range")

    def forecast(self, nout):
        o = _need()
        o.orc_mmcc_forecast.argtypes = [C.c_void_p, C.c_int]
        return o.orc_mmcc_forecast(self.s, nout)

    def general_work(self, nout, x, want_error=False):
        """returns (out[:n], err[:n] or None, consumed)"""
        o = _need()
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros(max(nout, 1), dtype=np.complex64)
        err = np.zeros(max(nout, 1), dtype=np.float32) if want_error else None
        consumed = C.c_int(0)
        o.orc_mmcc_general_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_int)]
        n = o.orc_mmcc_general_work(self.s, nout, len(x), x.ctypes.data, out.ctypes.data,
                                    err.ctypes.data if want_error else None, C.byref(consumed))
        return out[:n].copy(), (err[:n].copy() if want_error else None), consumed.value

    def mu(self):
        o = _need()
        o.orc_mmcc_mu.restype = C.c_float
        o.orc_mmcc_mu.argtypes = [C.c_void_p]
        return np.float32(o.orc_mmcc_mu(self.s))

    def omega(self):
        o = _need()
        o.orc_mmcc_omega.restype = C.c_float
        o.orc_mmcc_omega.argtypes = [C.c_void_p]
        return np.float32(o.orc_mmcc_omega(self.s))


class PagerSlicer:
    """pager_slicer_fb: state (d_avg) carried across work() calls"""

    def __init__(self, alpha):
        self.alpha = float(np.float32(alpha))
        self.avg = C.c_float(0.0)

    def work(self, x):
        o = _need()
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(len(x), dtype=np.uint8)
        o.orc_pager_slicer_fb.argtypes = [C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_size_t]
        o.orc_pager_slicer_fb(self.alpha, C.byref(self.avg), x.ctypes.data, out.ctypes.data, len(x))
        return out

    def dc_offset(self):
        return np.float32(self.avg.value)


class FramerSink1:
    """gr_framer_sink_1: state carried across work() calls; work() returns the messages
    [(whitener_offset, payload bytes), ...] the reference would have queued during the call"""

    def __init__(self):
        o = _need()
        o.orc_framer_state_size.restype = C.c_size_t
        self.state = C.create_string_buffer(o.orc_framer_state_size())
        o.orc_framer_sink_1_init.argtypes = [C.c_void_p]
        o.orc_framer_sink_1_init(self.state)

    def work(self, x):
        o = _need()
        x = np.ascontiguousarray(x, dtype=np.uint8)
        cap = len(x) // 32 + 2
        woff = np.zeros(cap, dtype=np.int32)
        mlen = np.zeros(cap, dtype=np.int32)
        pool = np.zeros(len(x) // 8 + 4096, dtype=np.uint8)
        used = C.c_size_t(0)
        o.orc_framer_sink_1_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_size_t)]
        o.orc_framer_sink_1_work.restype = C.c_int
        n = o.orc_framer_sink_1_work(self.state, x.ctypes.data, len(x), woff.ctypes.data, mlen.ctypes.data,
                                     pool.ctypes.data, C.byref(used))
        out, off = [], 0
        for i in range(n):
            out.append((int(woff[i]), pool[off:off + mlen[i]].tobytes()))
            off += int(mlen[i])
        return out


def unpack_k_bits_bb(k, x):
    o = _need()
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.zeros(len(x) * k, dtype=np.uint8)
    o.orc_unpack_k_bits_bb.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t]
    o.orc_unpack_k_bits_bb(int(k), x.ctypes.data, out.ctypes.data, len(out))
    return out


def binary_slicer_fb(x):
    o = _need()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(len(x), dtype=np.uint8)
    o.orc_binary_slicer_fb.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    o.orc_binary_slicer_fb(x.ctypes.data, out.ctypes.data, len(x))
    return out


def count_bits64(x):
    o = _need()
    o.orc_count_bits64.restype = C.c_uint
    o.orc_count_bits64.argtypes = [C.c_ulonglong]
    return np.array([o.orc_count_bits64(int(v)) for v in np.ravel(x)], dtype=np.uint32)


class _Corr(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("access_code", "data_reg", "flag_reg", "flag_bit", "mask")] + \
               [("threshold", C.c_uint)]


class CorrelateAccessCode:
    """digital_correlate_access_code_bb restatement."""

    def __init__(self, access_code, threshold):
        o = _need()
        self.c = _Corr()
        code = access_code.encode("latin-1") if isinstance(access_code, str) else bytes(access_code)
        o.orc_corr_init.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_int]
        if o.orc_corr_init(C.byref(self.c), code, len(code), int(threshold)):
            raise IndexError("access_code is > 64 bits")

    def work(self, bits):
        o = _need()
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        out = np.zeros(len(b), dtype=np.uint8)
        o.orc_corr_work.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        o.orc_corr_work(C.byref(self.c), b.ctypes.data, out.ctypes.data, len(b))
        return out


def fft_vcc(fft_size, forward, window, shift, x):
    o = _need()
    x = np.ascontiguousarray(x, dtype=np.complex64)
    nvec = len(x) // fft_size
    w = np.ascontiguousarray(window if window is not None else [], dtype=np.float32)
    out = np.zeros(nvec * fft_size, dtype=np.complex64)
    o.orc_fft_vcc.argtypes = [C.c_uint, C.c_int, C.c_void_p, C.c_uint, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_size_t]
    o.orc_fft_vcc(fft_size, int(bool(forward)), w.ctypes.data if len(w) else None, len(w),
                  int(bool(shift)), x.ctypes.data, out.ctypes.data, nvec)
    return out


class FftFilterCcc:
    """gr_fft_filter_ccc / gri_fft_filter_ccc_generic restatement (overlap-add)"""

    def __init__(self, decimation, taps):
        o = _need()
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        o.orc_fft_filter_new.restype = C.c_void_p
        o.orc_fft_filter_new.argtypes = [C.c_int, C.c_void_p, C.c_uint]
        self.h = o.orc_fft_filter_new(int(decimation), t.ctypes.data, len(t))
        if not self.h:
            raise ValueError("bad fft_filter arguments")
        o.orc_fft_filter_nsamples.argtypes = [C.c_void_p]
        self.nsamples = o.orc_fft_filter_nsamples(self.h)
        self.decim = int(decimation)

    def __del__(self):
        try:
            o = _need()
            o.orc_fft_filter_free.argtypes = [C.c_void_p]
            if self.h:
                o.orc_fft_filter_free(self.h)
                self.h = None
        except Exception:
            pass

    def filter(self, nitems, x):
        o = _need()
        x = np.ascontiguousarray(x, dtype=np.complex64)
        assert len(x) >= nitems * self.decim and (nitems * self.decim) % self.nsamples == 0
        out = np.zeros(nitems, dtype=np.complex64)
        o.orc_fft_filter_filter.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        o.orc_fft_filter_filter(self.h, int(nitems), x.ctypes.data, out.ctypes.data)
        return out


def pfb_decimator_ccf(decim, taps, chan, streams, n):
    """gr_pfb_decimator_ccf::work: streams[s] has taps_per_filter-1 old items in front"""
    o = _need()
    t = np.ascontiguousarray(taps, dtype=np.float32)
    arrs = [np.ascontiguousarray(s, dtype=np.complex64) for s in streams]
    ptrs = (C.c_void_p * decim)(*[a.ctypes.data for a in arrs])
    out = np.zeros(n, dtype=np.complex64)
    o.orc_pfb_decimator_ccf_work.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_int]
    o.orc_pfb_decimator_ccf_work(decim, t.ctypes.data, len(t), chan, ptrs, out.ctypes.data, n)
    return out


class PfbChannelizer:
    """gr_pfb_channelizer_ccf restatement."""

    def __init__(self, numchans, taps, oversample_rate=1.0):
        o = _need()
        t = np.ascontiguousarray(taps, dtype=np.float32)
        o.orc_pfb_new.restype = C.c_void_p
        o.orc_pfb_new.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_float]
        self.h = o.orc_pfb_new(numchans, t.ctypes.data, len(t), oversample_rate)
        if not self.h:
            raise ValueError("gr_pfb_channelizer: oversample rate must be N/i for i in [1, N]")
        self.M = numchans
        o.orc_pfb_taps_per_filter.argtypes = [C.c_void_p]
        self.taps_per_filter = o.orc_pfb_taps_per_filter(self.h)
        o.orc_pfb_output_multiple.argtypes = [C.c_void_p]
        self.output_multiple = o.orc_pfb_output_multiple(self.h)

    def general_work(self, nout, streams_with_history):
        """streams_with_history: list of M complex arrays, each with taps_per_filter
        history items in front.  Returns (out[nout, M], consumed)."""
        o = _need()
        arrs = [np.ascontiguousarray(s, dtype=np.complex64) for s in streams_with_history]
        ptrs = (C.c_void_p * self.M)(*[a.ctypes.data for a in arrs])
        out = np.zeros((nout, self.M), dtype=np.complex64)
        o.orc_pfb_general_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        consumed = o.orc_pfb_general_work(self.h, nout, ptrs, out.ctypes.data)
        return out, consumed

    def __del__(self):
        try:
            _orc.orc_pfb_free.argtypes = [C.c_void_p]
            _orc.orc_pfb_free(self.h)
        except Exception:
            pass


def chain_xlating_demod(decim, proto, center_freq, sampling_freq, gain, x, want_y=False, lib="oracle"):
    """Whole-capture chain: zeros history + xlating_ccc + quad_demod.
    lib = "oracle" (port) or "ref" (reference asm/headers)."""
    L = _need() if lib == "oracle" else _ref
    if L is None:
        raise RuntimeError("oracle/_ref/libgrref.so not present")
    fn = L.orc_chain_xlating_demod if lib == "oracle" else L.ref_chain_xlating_demod
    proto = np.ascontiguousarray(proto, dtype=np.complex64)
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n_out = len(x) // decim
    y = np.zeros(n_out, dtype=np.complex64) if want_y else None
    d = np.zeros(n_out, dtype=np.float32)
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_uint, C.c_void_p, C.c_uint, C.c_double, C.c_double, C.c_float, C.c_void_p,
                   C.c_size_t, C.c_void_p, C.c_void_p]
    fn(decim, proto.ctypes.data, len(proto), float(center_freq), float(sampling_freq),
       np.float32(gain), x.ctypes.data, len(x), y.ctypes.data if want_y else None, d.ctypes.data)
    return (y, d) if want_y else d


def chain_mm(omega, gain_omega, mu, gain_mu, rel_limit, x):
    o = _need()
    x = np.ascontiguousarray(x, dtype=np.float32)
    cap = int(len(x) / max(omega * (1 - rel_limit) - 0.5, 0.5)) + 16
    out = np.zeros(cap, dtype=np.float32)
    fs = np.zeros(4, dtype=np.float32)
    o.orc_chain_mm.argtypes = [C.c_float] * 5 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = o.orc_chain_mm(omega, gain_omega, mu, gain_mu, rel_limit, x.ctypes.data, len(x),
                       out.ctypes.data, cap, fs.ctypes.data)
    if n < 0:
        raise IndexError("out_of_range")
    return out[:n].copy(), dict(mu=fs[0], omega=fs[1], last_sample=fs[2], consumed=int(fs[3]))


# ----------------------------------------------------------------------------
# reference pieces (oracle/_ref/libgrref.so) -- only where buildable
# ----------------------------------------------------------------------------
def ref_fast_atan2f(y, x):
    y = np.ascontiguousarray(y, dtype=np.float32).ravel()
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    out = np.empty_like(y)
    _ref.ref_fast_atan2f_n.argtypes = [_f32p, _f32p, _f32p, C.c_size_t]
    _ref.ref_fast_atan2f_n(y, x, out, len(y))
    return out


def ref_rotator_phases(incr, n):
    out = np.zeros(n, dtype=np.complex64)
    _ref.ref_rotator_phases.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_size_t]
    _ref.ref_rotator_phases(np.float32(incr.real), np.float32(incr.imag), out.ctypes.data, n)
    return out


def ref_branchless_clip(x, clip):
    x = np.ascontiguousarray(x, dtype=np.float32).ravel()
    out = np.empty_like(x)
    _ref.ref_branchless_clip_n.argtypes = [_f32p, C.c_float, _f32p, C.c_size_t]
    _ref.ref_branchless_clip_n(x, np.float32(clip), out, len(x))
    return out


def ref_count_bits64(x):
    x = np.ascontiguousarray(x, dtype=np.uint64).ravel()
    out = np.zeros(len(x), dtype=np.uint32)
    _ref.ref_count_bits64_n.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _ref.ref_count_bits64_n(x.ctypes.data, out.ctypes.data, len(x))
    return out


def ref_mmse_taps():
    out = np.zeros((129, 8), dtype=np.float32)
    _ref.ref_mmse_taps.argtypes = [C.c_void_p]
    _ref.ref_mmse_taps(out.ctypes.data)
    return out


def _aligned(arr, front_pad_items, dtype):
    """copy into a 16B-aligned buffer with slack both sides (the SSE path reads
    below `input` down to a 16-byte boundary and past the end)."""
    a = np.ascontiguousarray(arr, dtype=dtype)
    isz = a.dtype.itemsize
    raw = np.zeros(len(a) * isz + 256, dtype=np.uint8)
    off = (-raw.ctypes.data) % 16 + 64 + front_pad_items * isz
    view = raw[off:off + len(a) * isz].view(dtype)
    view[:] = a
    return raw, view


def ref_fir_sse(kind, taps, x, n, decim=1, misalign=0):
    """kind in {'fff','ccf','ccc'}; misalign shifts the input start by that
    many items relative to a 16-byte boundary."""
    tap_c = kind == "ccc"
    x_c = kind != "fff"
    taps = np.ascontiguousarray(taps, dtype=np.complex64 if tap_c else np.float32)
    raw, xv = _aligned(x, misalign, np.complex64 if x_c else np.float32)
    out = np.zeros(n, dtype=np.complex64 if x_c else np.float32)
    fn = getattr(_ref, "ref_fir_%s_sse" % kind)
    fn.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]
    fn(taps.ctypes.data, len(taps), xv.ctypes.data, out.ctypes.data, n, decim)
    return out
