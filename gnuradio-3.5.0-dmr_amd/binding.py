"""ctypes binding of libgrhip.so.  See include/grhip.h for the contract."""
import ctypes as C
import os

import numpy as np

__all__ = [
    "GrhipError", "lib", "lib_path", "strerror", "device_count", "set_default_mode",
    "MODE_FAST", "MODE_GENERIC", "MODE_FAST_VALU", "MODE_FAST_REFTAPS", "WORK_DONE",
    "fir_filter_ccf", "fir_filter_fff", "fir_filter_ccc", "fir_filter_with_buffer",
    "freq_xlating_fir_filter_ccc", "quadrature_demod_cf", "xlating_demod",
    "clock_recovery_mm_ff", "clock_recovery_mm_cc", "binary_slicer_fb", "correlate_access_code_bb", "pager_slicer_fb", "unpack_k_bits_bb", "framer_sink_1", "framer_sink_1_batch", "stream_to_streams", "streams_to_stream", "vector_to_streams", "stream_to_vector", "head",
    "fft_vcc", "fft_filter_ccc", "pfb_channelizer_ccf", "pfb_decimator_ccf", "dmr_chain", "run_sync_block",
]

MODE_FAST = 0
MODE_GENERIC = 1
MODE_FAST_VALU = 2     # FAST without the matrix cores (vector FMAs only)
MODE_FAST_REFTAPS = 3  # FAST + the reference's tap-angle quantisation reproduced by freq_xlating's matrix-core engine
WORK_DONE = 0x7fffffff  # GRHIP_WORK_DONE: not an error, not an item count

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class GrhipError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        RuntimeError.__init__(self, "grhip error %d (%s): %s" % (code, strerror(code), detail))


def lib_path():
    # GRHIP_LIB: load another build of the same library (diagnostic builds only)
    return os.environ.get("GRHIP_LIB") or os.path.join(_HERE, "libgrhip.so")


def lib():
    """Load libgrhip.so (built in-tree by `make -C gnuradio-3.5.0-dmr_amd` /
    __graft_entry__.build()).  Fails loudly when it is missing."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64
        # with the same SONAME.  If torch is going to be used for device memory /
        # streams / torch.distributed, it must be the copy that gets loaded first,
        # otherwise torch later reports "No HIP GPUs are available".
        if os.environ.get("GRHIP_NO_TORCH_PRELOAD") is None:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(p):
            raise ImportError("libgrhip.so not built (%s): run __graft_entry__.build(); "
                              "there is no CPU fallback" % p)
        _LIB = C.CDLL(p)
        _LIB.grhip_strerror.restype = C.c_char_p
        _LIB.grhip_last_error.restype = C.c_char_p
        _LIB.grhip_version.restype = C.c_char_p
        for n in ("grhip_clock_recovery_mm_ff_mu", "grhip_clock_recovery_mm_ff_omega",
                  "grhip_clock_recovery_mm_ff_gain_mu", "grhip_clock_recovery_mm_ff_gain_omega"):
            getattr(_LIB, n).restype = C.c_float
    return _LIB


def strerror(code):
    try:
        return lib().grhip_strerror(int(code)).decode()
    except Exception:
        return "?"


def _check(rc):
    if rc < 0:
        raise GrhipError(rc, lib().grhip_last_error().decode())
    return rc


def _raise_like_reference(rc):
    """map status codes to the exception types the reference's SWIG layer
    surfaces for the same precondition (RuntimeError via %exception,
    gnuradio-core/src/lib/swig/gnuradio.i:33-44)."""
    if rc < 0:
        raise GrhipError(rc, lib().grhip_last_error().decode())
    return rc


def device_count():
    n = C.c_int(0)
    rc = lib().grhip_device_count(C.byref(n))
    if rc < 0:
        return 0
    return n.value


def set_default_mode(mode):
    _check(lib().grhip_set_default_mode(int(mode)))


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _devptr(t):
    """accept an int address or anything with data_ptr() (torch tensor)"""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


def _stream(s):
    if s is None:
        return C.c_void_p(0)
    if hasattr(s, "cuda_stream"):
        return C.c_void_p(s.cuda_stream)
    return C.c_void_p(int(s))


class _Block(object):
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p(0)

    def __del__(self):
        try:
            if self._h and self._destroy:
                getattr(lib(), self._destroy)(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass


# ----------------------------------------------------------------------------
# gr.fir_filter_XXX  (filter/gr_fir_filter_XXX.i.t:28-41)
# ----------------------------------------------------------------------------
class _fir_filter(_Block):
    _destroy = "grhip_fir_filter_destroy"
    _kind = None
    _in = np.complex64
    _out = np.complex64
    _tap = np.float32

    def __init__(self, decimation, taps, device=0):
        _Block.__init__(self)
        t = np.ascontiguousarray(taps, dtype=self._tap)
        L = lib()
        L.grhip_fir_filter_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_void_p,
                                              C.c_size_t, C.c_int]
        _check(L.grhip_fir_filter_create(C.byref(self._h), self._kind.encode(), int(decimation),
                                         _ptr(t), len(t), int(device)))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=self._tap)
        L = lib()
        L.grhip_fir_filter_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_fir_filter_set_taps(self._h, _ptr(t), len(t)))

    def set_mode(self, mode):
        _check(lib().grhip_fir_filter_set_mode(self._h, int(mode)))

    def history(self):
        return _check(lib().grhip_fir_filter_history(self._h))

    def decimation(self):
        return _check(lib().grhip_fir_filter_decimation(self._h))

    def work(self, noutput_items, input_items):
        """input_items: numpy array with history()-1 old items in front."""
        x = np.ascontiguousarray(input_items, dtype=self._in)
        need = noutput_items * self.decimation() + self.history() - 1
        if len(x) < need:
            raise ValueError("work needs %d input items, got %d" % (need, len(x)))
        out = np.zeros(noutput_items, dtype=self._out)
        L = lib()
        L.grhip_fir_filter_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_fir_filter_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_fir_filter_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_fir_filter_work_device(self._h, int(noutput_items), _devptr(d_in),
                                                     _devptr(d_out), _stream(stream)))

    def filterNdec(self, x, n, decimate):
        x = np.ascontiguousarray(x, dtype=self._in)
        out = np.zeros(n, dtype=self._out)
        L = lib()
        L.grhip_fir_filterNdec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong, C.c_uint]
        _check(L.grhip_fir_filterNdec(self._h, _ptr(out), _ptr(x), n, decimate))
        return out


class fir_filter_with_buffer(_Block):
    """gri_fir_filter_with_buffer_{ccf,ccc,fff}: the FIR kernel object that keeps its own delay line"""
    _destroy = "grhip_fir_filter_with_buffer_destroy"

    def __init__(self, kind, taps, device=0):
        _Block.__init__(self)
        self.kind = kind
        self._tap = np.complex64 if kind == "ccc" else np.float32
        self._io = np.float32 if kind == "fff" else np.complex64
        t = np.ascontiguousarray(taps, dtype=self._tap)
        L = lib()
        L.grhip_fir_filter_with_buffer_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_void_p, C.c_size_t, C.c_int]
        _check(L.grhip_fir_filter_with_buffer_create(C.byref(self._h), kind.encode(), _ptr(t), len(t), int(device)))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=self._tap)
        L = lib()
        L.grhip_fir_filter_with_buffer_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_fir_filter_with_buffer_set_taps(self._h, _ptr(t), len(t)))

    def set_mode(self, mode):
        _check(lib().grhip_fir_filter_with_buffer_set_mode(self._h, int(mode)))

    def ntaps(self):
        return _check(lib().grhip_fir_filter_with_buffer_ntaps(self._h))

    def filterNdec(self, x, n, decimate=1):
        x = np.ascontiguousarray(x, dtype=self._io)
        if len(x) < n * decimate:
            raise ValueError("filterNdec needs %d items" % (n * decimate))
        out = np.zeros(n, dtype=self._io)
        L = lib()
        L.grhip_fir_filter_with_buffer_filterNdec.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong, C.c_ulong]
        _check(L.grhip_fir_filter_with_buffer_filterNdec(self._h, _ptr(out), _ptr(x), n, decimate))
        return out

    def filterN(self, x, n):
        return self.filterNdec(x, n, 1)

    def filterNdec_device(self, d_out, d_in, n, decimate=1, stream=None):
        L = lib()
        L.grhip_fir_filter_with_buffer_filterNdec_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong,
                                                                     C.c_ulong, C.c_void_p]
        _check(L.grhip_fir_filter_with_buffer_filterNdec_device(self._h, _devptr(d_out), _devptr(d_in), n, decimate,
                                                                _stream(stream)))


class fir_filter_ccf(_fir_filter):
    _kind = "ccf"


class fir_filter_fff(_fir_filter):
    _kind = "fff"
    _in = np.float32
    _out = np.float32


class fir_filter_ccc(_fir_filter):
    _kind = "ccc"
    _tap = np.complex64


# ----------------------------------------------------------------------------
# gr.freq_xlating_fir_filter_ccc
# ----------------------------------------------------------------------------
class freq_xlating_fir_filter_ccc(_Block):
    _destroy = "grhip_freq_xlating_fir_filter_ccc_destroy"

    def __init__(self, decimation, taps, center_freq, sampling_freq, device=0):
        _Block.__init__(self)
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        self._decim = int(decimation)
        L = lib()
        L.grhip_freq_xlating_fir_filter_ccc_create.argtypes = [
            C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int]
        _check(L.grhip_freq_xlating_fir_filter_ccc_create(C.byref(self._h), self._decim, _ptr(t), len(t),
                                                          float(center_freq), float(sampling_freq),
                                                          int(device)))

    def set_center_freq(self, center_freq):
        L = lib()
        L.grhip_freq_xlating_fir_filter_ccc_set_center_freq.argtypes = [C.c_void_p, C.c_double]
        _check(L.grhip_freq_xlating_fir_filter_ccc_set_center_freq(self._h, float(center_freq)))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        L = lib()
        L.grhip_freq_xlating_fir_filter_ccc_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_freq_xlating_fir_filter_ccc_set_taps(self._h, _ptr(t), len(t)))

    def set_mode(self, mode):
        _check(lib().grhip_freq_xlating_fir_filter_ccc_set_mode(self._h, int(mode)))

    def reset(self):
        _check(lib().grhip_freq_xlating_fir_filter_ccc_reset(self._h))

    def history(self):
        return _check(lib().grhip_freq_xlating_fir_filter_ccc_history(self._h))

    def decimation(self):
        return self._decim

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.complex64)
        need = noutput_items * self._decim + self.history() - 1
        if len(x) < need:
            raise ValueError("work needs %d input items, got %d" % (need, len(x)))
        out = np.zeros(noutput_items, dtype=np.complex64)
        L = lib()
        L.grhip_freq_xlating_fir_filter_ccc_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_freq_xlating_fir_filter_ccc_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_freq_xlating_fir_filter_ccc_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                                                    C.c_void_p, C.c_void_p]
        return _check(L.grhip_freq_xlating_fir_filter_ccc_work_device(
            self._h, int(noutput_items), _devptr(d_in), _devptr(d_out), _stream(stream)))


# ----------------------------------------------------------------------------
# gr.quadrature_demod_cf
# ----------------------------------------------------------------------------
class quadrature_demod_cf(_Block):
    _destroy = "grhip_quadrature_demod_cf_destroy"

    def __init__(self, gain, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_quadrature_demod_cf_create.argtypes = [C.POINTER(C.c_void_p), C.c_float, C.c_int]
        _check(L.grhip_quadrature_demod_cf_create(C.byref(self._h), float(gain), int(device)))

    def history(self):
        return 2

    def decimation(self):
        return 1

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.complex64)
        if len(x) < noutput_items + 1:
            raise ValueError("work needs %d input items" % (noutput_items + 1))
        out = np.zeros(noutput_items, dtype=np.float32)
        L = lib()
        L.grhip_quadrature_demod_cf_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_quadrature_demod_cf_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_quadrature_demod_cf_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                            C.c_void_p]
        return _check(L.grhip_quadrature_demod_cf_work_device(self._h, int(noutput_items), _devptr(d_in),
                                                              _devptr(d_out), _stream(stream)))


# ----------------------------------------------------------------------------
# fused hier block xlating -> quad_demod
# ----------------------------------------------------------------------------
class xlating_demod(_Block):
    _destroy = "grhip_xlating_demod_destroy"

    def __init__(self, decimation, taps, center_freq, sampling_freq, gain, device=0):
        _Block.__init__(self)
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        self._decim = int(decimation)
        L = lib()
        L.grhip_xlating_demod_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t,
                                                 C.c_double, C.c_double, C.c_float, C.c_int]
        _check(L.grhip_xlating_demod_create(C.byref(self._h), self._decim, _ptr(t), len(t),
                                            float(center_freq), float(sampling_freq), float(gain),
                                            int(device)))

    def set_mode(self, mode):
        _check(lib().grhip_xlating_demod_set_mode(self._h, int(mode)))

    def reset(self):
        _check(lib().grhip_xlating_demod_reset(self._h))

    def history(self):
        return _check(lib().grhip_xlating_demod_history(self._h))

    def decimation(self):
        return self._decim

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.complex64)
        need = noutput_items * self._decim + self.history() - 1
        if len(x) < need:
            raise ValueError("work needs %d input items, got %d" % (need, len(x)))
        out = np.zeros(noutput_items, dtype=np.float32)
        L = lib()
        L.grhip_xlating_demod_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_xlating_demod_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_xlating_demod_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_xlating_demod_work_device(self._h, int(noutput_items), _devptr(d_in),
                                                        _devptr(d_out), _stream(stream)))

    def run_captures_device(self, n_streams, n_samples, d_in, in_stride_items, d_out, out_stride_items,
                            stream=None):
        """n_streams fresh-state captures (no history in front) in one launch"""
        L = lib()
        L.grhip_xlating_demod_run_captures_device.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p,
                                                              C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        return _check(L.grhip_xlating_demod_run_captures_device(
            self._h, int(n_streams), int(n_samples), _devptr(d_in), int(in_stride_items), _devptr(d_out),
            int(out_stride_items), _stream(stream)))


# ----------------------------------------------------------------------------
# digital.clock_recovery_mm_ff
# ----------------------------------------------------------------------------
class clock_recovery_mm_ff(_Block):
    _destroy = "grhip_clock_recovery_mm_ff_destroy"

    def __init__(self, omega, gain_omega, mu, gain_mu, omega_relative_limit, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_clock_recovery_mm_ff_create.argtypes = [C.POINTER(C.c_void_p)] + [C.c_float] * 5 + [C.c_int]
        _check(L.grhip_clock_recovery_mm_ff_create(C.byref(self._h), omega, gain_omega, mu, gain_mu,
                                                   omega_relative_limit, int(device)))

    def forecast(self, noutput_items):
        return _check(lib().grhip_clock_recovery_mm_ff_forecast(self._h, int(noutput_items)))

    def general_work(self, noutput_items, input_items):
        """returns (out, consumed)"""
        x = np.ascontiguousarray(input_items, dtype=np.float32)
        out = np.zeros(max(int(noutput_items), 1), dtype=np.float32)
        consumed = C.c_int(0)
        L = lib()
        L.grhip_clock_recovery_mm_ff_general_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                              C.c_void_p, C.POINTER(C.c_int)]
        n = _check(L.grhip_clock_recovery_mm_ff_general_work(self._h, int(noutput_items), len(x), _ptr(x),
                                                             _ptr(out), C.byref(consumed)))
        return out[:n].copy(), consumed.value

    def general_work_device(self, noutput_items, ninput_items, d_in, d_out, d_counts, stream=None):
        L = lib()
        L.grhip_clock_recovery_mm_ff_general_work_device.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_clock_recovery_mm_ff_general_work_device(
            self._h, int(noutput_items), int(ninput_items), _devptr(d_in), _devptr(d_out),
            _devptr(d_counts), _stream(stream)))

    def mu(self):
        return lib().grhip_clock_recovery_mm_ff_mu(self._h)

    def omega(self):
        return lib().grhip_clock_recovery_mm_ff_omega(self._h)

    def gain_mu(self):
        return lib().grhip_clock_recovery_mm_ff_gain_mu(self._h)

    def gain_omega(self):
        return lib().grhip_clock_recovery_mm_ff_gain_omega(self._h)

    def _setf(self, name, v):
        f = getattr(lib(), "grhip_clock_recovery_mm_ff_set_" + name)
        f.argtypes = [C.c_void_p, C.c_float]
        _check(f(self._h, float(v)))

    def set_gain_mu(self, v):
        self._setf("gain_mu", v)

    def set_gain_omega(self, v):
        self._setf("gain_omega", v)

    def set_mu(self, v):
        self._setf("mu", v)

    def set_omega(self, v):
        self._setf("omega", v)


# ----------------------------------------------------------------------------
# digital.binary_slicer_fb / digital.correlate_access_code_bb
# ----------------------------------------------------------------------------
class binary_slicer_fb(_Block):
    _destroy = "grhip_binary_slicer_fb_destroy"

    def __init__(self, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_binary_slicer_fb_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        _check(L.grhip_binary_slicer_fb_create(C.byref(self._h), int(device)))

    def history(self):
        return 1

    def decimation(self):
        return 1

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.float32)
        out = np.zeros(noutput_items, dtype=np.uint8)
        L = lib()
        L.grhip_binary_slicer_fb_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_binary_slicer_fb_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_binary_slicer_fb_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_binary_slicer_fb_work_device(self._h, int(noutput_items), _devptr(d_in), _devptr(d_out),
                                                           _stream(stream)))


class pager_slicer_fb(_Block):
    """pager.slicer_fb(alpha): DC-tracking 4-level slicer (gr-pager/lib/pager_slicer_fb.cc)"""
    _destroy = "grhip_pager_slicer_fb_destroy"

    def __init__(self, alpha, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_pager_slicer_fb_create.argtypes = [C.POINTER(C.c_void_p), C.c_float, C.c_int]
        _check(L.grhip_pager_slicer_fb_create(C.byref(self._h), float(alpha), int(device)))

    def history(self):
        return 1

    def decimation(self):
        return 1

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.float32)
        out = np.zeros(noutput_items, dtype=np.uint8)
        L = lib()
        L.grhip_pager_slicer_fb_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_pager_slicer_fb_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_pager_slicer_fb_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_pager_slicer_fb_work_device(self._h, int(noutput_items), _devptr(d_in), _devptr(d_out),
                                                          _stream(stream)))

    def dc_offset(self):
        L = lib()
        v = C.c_float(0)
        L.grhip_pager_slicer_fb_dc_offset.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _check(L.grhip_pager_slicer_fb_dc_offset(self._h, C.byref(v)))
        return np.float32(v.value)


class unpack_k_bits_bb(_Block):
    """gr.unpack_k_bits_bb(k): k output bytes (one bit each, MSB first) per input byte"""
    _destroy = "grhip_unpack_k_bits_bb_destroy"

    def __init__(self, k, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_unpack_k_bits_bb_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
        _check(L.grhip_unpack_k_bits_bb_create(C.byref(self._h), int(k), int(device)))
        self.k = int(k)

    def interpolation(self):
        return self.k

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.uint8)
        out = np.zeros(noutput_items, dtype=np.uint8)
        L = lib()
        L.grhip_unpack_k_bits_bb_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_unpack_k_bits_bb_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]


class clock_recovery_mm_cc(_Block):
    """digital.clock_recovery_mm_cc(omega, gain_omega, mu, gain_mu, omega_relative_limit)"""
    _destroy = "grhip_clock_recovery_mm_cc_destroy"

    def __init__(self, omega, gain_omega, mu, gain_mu, omega_relative_limit, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_clock_recovery_mm_cc_create.argtypes = [C.POINTER(C.c_void_p)] + [C.c_float] * 5 + [C.c_int]
        _raise_like_reference(L.grhip_clock_recovery_mm_cc_create(C.byref(self._h), omega, gain_omega, mu, gain_mu,
                                                                  omega_relative_limit, int(device)))

    def forecast(self, noutput_items):
        return _check(lib().grhip_clock_recovery_mm_cc_forecast(self._h, int(noutput_items)))

    def history(self):
        return _check(lib().grhip_clock_recovery_mm_cc_history(self._h))

    def _get(self, name):
        v = C.c_float(0)
        f = getattr(lib(), "grhip_clock_recovery_mm_cc_" + name)
        f.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        _check(f(self._h, C.byref(v)))
        return np.float32(v.value)

    def _set(self, name, v):
        f = getattr(lib(), "grhip_clock_recovery_mm_cc_set_" + name)
        f.argtypes = [C.c_void_p, C.c_float]
        _check(f(self._h, float(v)))

    def mu(self): return self._get("mu")
    def omega(self): return self._get("omega")
    def gain_mu(self): return self._get("gain_mu")
    def gain_omega(self): return self._get("gain_omega")
    def set_mu(self, v): self._set("mu", v)
    def set_omega(self, v): self._set("omega", v)
    def set_gain_mu(self, v): self._set("gain_mu", v)
    def set_gain_omega(self, v): self._set("gain_omega", v)

    def general_work(self, noutput_items, input_items, want_error=False):
        """returns (out[:n], err[:n] or None, consumed)"""
        x = np.ascontiguousarray(input_items, dtype=np.complex64)
        out = np.zeros(max(noutput_items, 1), dtype=np.complex64)
        err = np.zeros(max(noutput_items, 1), dtype=np.float32) if want_error else None
        consumed = C.c_int(0)
        L = lib()
        L.grhip_clock_recovery_mm_cc_general_work.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                             C.c_void_p, C.POINTER(C.c_int)]
        n = _check(L.grhip_clock_recovery_mm_cc_general_work(self._h, int(noutput_items), len(x), _ptr(x), _ptr(out),
                                                             _ptr(err) if want_error else None, C.byref(consumed)))
        return out[:n], (err[:n] if want_error else None), consumed.value


class _copy_adapter(_Block):
    _destroy = "grhip_copy_adapter_destroy"

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items)
        out = np.zeros(max(noutput_items, 1) * self.out_bytes, dtype=np.uint8)
        L = lib()
        L.grhip_copy_adapter_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_copy_adapter_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))   # negative: always an error
        if n == WORK_DONE:
            return None
        return out[:n * self.out_bytes].view(x.dtype)

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_copy_adapter_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_copy_adapter_work_device(self._h, int(noutput_items), _devptr(d_in), _devptr(d_out),
                                                    _stream(stream)))
        return None if n == WORK_DONE else n


class stream_to_vector(_copy_adapter):
    """gr.stream_to_vector(item_size, nitems_per_block)"""

    def __init__(self, item_size, nitems_per_block, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_stream_to_vector_create.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_int]
        _check(L.grhip_stream_to_vector_create(C.byref(self._h), int(item_size), int(nitems_per_block), int(device)))
        self.out_bytes = int(item_size) * int(nitems_per_block)


class head(_copy_adapter):
    """gr.head(sizeof_stream_item, nitems): work() / work_device() return None (WORK_DONE) once nitems have passed"""

    def __init__(self, sizeof_stream_item, nitems, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_head_create.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_ulonglong, C.c_int]
        _check(L.grhip_head_create(C.byref(self._h), int(sizeof_stream_item), int(nitems), int(device)))
        self.out_bytes = int(sizeof_stream_item)

    def reset(self):
        _check(lib().grhip_head_reset(self._h))


class framer_sink_1_batch(_Block):
    """multi-capture gr.framer_sink_1: run_device() frames n_streams item streams in one go (every stream from
    the search state); messages() -> [[(whitener_offset, payload bytes), ...] per stream]"""
    _destroy = "grhip_framer_sink_1_batch_destroy"

    def __init__(self, n_streams, max_items_per_stream, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_framer_sink_1_batch_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int]
        _check(L.grhip_framer_sink_1_batch_create(C.byref(self._h), int(n_streams), int(max_items_per_stream), int(device)))
        self.n_streams = int(n_streams)

    def run_device(self, d_in, stream_stride_items, d_nitems, n_items_max, stream=None, nitems_stride=1):
        L = lib()
        L.grhip_framer_sink_1_batch_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int,
                                                           C.c_size_t, C.c_void_p]
        _check(L.grhip_framer_sink_1_batch_run_device(self._h, _devptr(d_in), int(stream_stride_items), _devptr(d_nitems),
                                                      int(nitems_stride), int(n_items_max), _stream(stream)))

    def messages(self, stream=None):
        L = lib()
        L.grhip_framer_sink_1_batch_fetch.argtypes = [C.c_void_p, C.c_void_p]
        L.grhip_framer_sink_1_batch_count.argtypes = [C.c_void_p, C.c_int]
        L.grhip_framer_sink_1_batch_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int]
        _check(L.grhip_framer_sink_1_batch_fetch(self._h, _stream(stream)))
        buf = np.zeros(4096, np.uint8)
        out = []
        for s in range(self.n_streams):
            msgs = []
            for i in range(_check(L.grhip_framer_sink_1_batch_count(self._h, s))):
                w = C.c_int(0)
                ln = _check(L.grhip_framer_sink_1_batch_get(self._h, s, i, C.byref(w), _ptr(buf), 4096))
                msgs.append((w.value, buf[:ln].tobytes()))
            out.append(msgs)
        return out


class framer_sink_1(_Block):
    """gr.framer_sink_1(msgq): header + payload extraction after the correlator's flag bit.
    The reference inserts gr.message objects into `msgq`; here work() / work_device() collect them
    and messages() returns (and removes) [(whitener_offset, payload bytes), ...] in order."""
    _destroy = "grhip_framer_sink_1_destroy"

    def __init__(self, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_framer_sink_1_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        _check(L.grhip_framer_sink_1_create(C.byref(self._h), int(device)))

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.uint8)
        L = lib()
        L.grhip_framer_sink_1_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        return _check(L.grhip_framer_sink_1_work(self._h, int(noutput_items), _ptr(x)))

    def set_segment_items(self, items):
        """tuning: items per segment of the parallel walk of long calls (0 = automatic); results do not depend on it"""
        L = lib()
        L.grhip_framer_sink_1_set_segment_items.argtypes = [C.c_void_p, C.c_longlong]
        _check(L.grhip_framer_sink_1_set_segment_items(self._h, int(items)))

    def work_device(self, noutput_items, d_in, stream=None):
        L = lib()
        L.grhip_framer_sink_1_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        return _check(L.grhip_framer_sink_1_work_device(self._h, int(noutput_items), _devptr(d_in), _stream(stream)))

    def messages(self, stream=None):
        L = lib()
        L.grhip_framer_sink_1_message_count.argtypes = [C.c_void_p, C.c_void_p]
        L.grhip_framer_sink_1_drain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        n = _check(L.grhip_framer_sink_1_message_count(self._h, _stream(stream)))
        out = []
        while n > 0:
            k = min(n, 1 << 16)
            woff = np.zeros(k, dtype=np.int32)
            lens = np.zeros(k, dtype=np.int32)
            buf = np.zeros(k * 4096 if k < 64 else max(k * 256, 1 << 20), dtype=np.uint8)
            got = _check(L.grhip_framer_sink_1_drain(self._h, k, _ptr(woff), _ptr(lens), _ptr(buf), buf.size))
            if got == 0:                                   # a single payload larger than the share of the buffer
                buf = np.zeros(4096, dtype=np.uint8)
                got = _check(L.grhip_framer_sink_1_drain(self._h, 1, _ptr(woff), _ptr(lens), _ptr(buf), buf.size))
            ends = np.cumsum(lens[:got])
            raw = buf.tobytes()
            for i in range(got):
                out.append((int(woff[i]), raw[ends[i] - lens[i]:ends[i]]))
            n -= got
        return out


class _stream_adapter(_Block):
    _destroy = "grhip_stream_adapter_destroy"
    _split = 1

    def __init__(self, item_size, nstreams, device=0):
        _Block.__init__(self)
        L = lib()
        L.grhip_stream_adapter_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_size_t, C.c_int]
        _check(L.grhip_stream_adapter_create(C.byref(self._h), self._split, int(item_size), int(nstreams), int(device)))
        self.item_size, self.nstreams = int(item_size), int(nstreams)

    def _work(self, n, single, streams):
        L = lib()
        arr = (C.c_void_p * self.nstreams)(*[s.ctypes.data for s in streams])
        L.grhip_stream_adapter_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        return _check(L.grhip_stream_adapter_work(self._h, int(n), single.ctypes.data, arr))

    def work_device(self, n_items_per_stream, d_single, d_streams, stream_stride_items, stream=None):
        L = lib()
        L.grhip_stream_adapter_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        return _check(L.grhip_stream_adapter_work_device(self._h, int(n_items_per_stream), _devptr(d_single),
                                                         _devptr(d_streams), int(stream_stride_items), _stream(stream)))


class stream_to_streams(_stream_adapter):
    """gr.stream_to_streams(item_size, nstreams): work(noutput_items, input) -> list of nstreams arrays"""
    _split = 1

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items)
        assert x.dtype.itemsize == self.item_size
        outs = [np.zeros(noutput_items, dtype=x.dtype) for _ in range(self.nstreams)]
        self._work(noutput_items, x, outs)
        return outs


class vector_to_streams(stream_to_streams):
    """gr.vector_to_streams(item_size, nstreams): item j of every input vector -> stream j (same data movement as
    stream_to_streams, general/gr_vector_to_streams.cc:45-70)"""


class streams_to_stream(_stream_adapter):
    """gr.streams_to_stream(item_size, nstreams): work(noutput_items, [inputs]) -> one array"""
    _split = 0

    def work(self, noutput_items, input_items):
        ins = [np.ascontiguousarray(a) for a in input_items]
        assert noutput_items % self.nstreams == 0 and ins[0].dtype.itemsize == self.item_size
        out = np.zeros(noutput_items, dtype=ins[0].dtype)
        self._work(noutput_items // self.nstreams, out, ins)
        return out


class correlate_access_code_bb(_Block):
    _destroy = "grhip_correlate_access_code_bb_destroy"

    def __init__(self, access_code, threshold, device=0):
        _Block.__init__(self)
        code = access_code.encode("latin-1") if isinstance(access_code, str) else bytes(access_code)
        L = lib()
        L.grhip_correlate_access_code_bb_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t,
                                                            C.c_int, C.c_int]
        _check(L.grhip_correlate_access_code_bb_create(C.byref(self._h), code, len(code), int(threshold),
                                                       int(device)))

    def set_access_code(self, access_code):
        code = access_code.encode("latin-1") if isinstance(access_code, str) else bytes(access_code)
        L = lib()
        L.grhip_correlate_access_code_bb_set_access_code.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        rc = L.grhip_correlate_access_code_bb_set_access_code(self._h, code, len(code))
        return rc == 0   # bool like the reference (digital_correlate_access_code_bb.cc:64-68)

    def history(self):
        return 1

    def decimation(self):
        return 1

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.uint8)
        out = np.zeros(noutput_items, dtype=np.uint8)
        L = lib()
        L.grhip_correlate_access_code_bb_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_correlate_access_code_bb_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_correlate_access_code_bb_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                                 C.c_void_p]
        return _check(L.grhip_correlate_access_code_bb_work_device(
            self._h, int(noutput_items), _devptr(d_in), _devptr(d_out), _stream(stream)))


# ----------------------------------------------------------------------------
# gr.fft_vcc
# ----------------------------------------------------------------------------
class fft_vcc(_Block):
    _destroy = "grhip_fft_vcc_destroy"

    def __init__(self, fft_size, forward, window, shift=False, device=0):
        _Block.__init__(self)
        w = np.ascontiguousarray(window if window is not None else [], dtype=np.float32)
        self.fft_size = int(fft_size)
        L = lib()
        L.grhip_fft_vcc_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                           C.c_int, C.c_int]
        _check(L.grhip_fft_vcc_create(C.byref(self._h), self.fft_size, int(bool(forward)),
                                      _ptr(w) if len(w) else None, len(w), int(bool(shift)), int(device)))

    def set_window(self, window):
        w = np.ascontiguousarray(window, dtype=np.float32)
        L = lib()
        L.grhip_fft_vcc_set_window.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        return bool(_check(L.grhip_fft_vcc_set_window(self._h, _ptr(w) if len(w) else None, len(w))))

    def work(self, noutput_items, input_items):
        """items are vectors: input_items has noutput_items*fft_size complex."""
        x = np.ascontiguousarray(input_items, dtype=np.complex64).reshape(-1)
        if len(x) < noutput_items * self.fft_size:
            raise ValueError("not enough input")
        out = np.zeros(noutput_items * self.fft_size, dtype=np.complex64)
        L = lib()
        L.grhip_fft_vcc_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_fft_vcc_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n * self.fft_size]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_fft_vcc_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_fft_vcc_work_device(self._h, int(noutput_items), _devptr(d_in),
                                                  _devptr(d_out), _stream(stream)))


# ----------------------------------------------------------------------------
# gr.pfb_channelizer_ccf
# ----------------------------------------------------------------------------
class fft_filter_ccc(_Block):
    """gr.fft_filter_ccc(decimation, taps): overlap-add fast convolution (history 1, output multiple nsamples)"""
    _destroy = "grhip_fft_filter_ccc_destroy"

    def __init__(self, decimation, taps, device=0):
        _Block.__init__(self)
        L = lib()
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        L.grhip_fft_filter_ccc_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        _check(L.grhip_fft_filter_ccc_create(C.byref(self._h), int(decimation), _ptr(t), len(t), int(device)))

    def history(self):
        return 1

    def decimation(self):
        return _check(lib().grhip_fft_filter_ccc_decimation(self._h))

    def nsamples(self):
        """the block's output multiple"""
        return _check(lib().grhip_fft_filter_ccc_nsamples(self._h))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.complex64)
        L = lib()
        L.grhip_fft_filter_ccc_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_fft_filter_ccc_set_taps(self._h, _ptr(t), len(t)))

    def work(self, noutput_items, input_items):
        x = np.ascontiguousarray(input_items, dtype=np.complex64)
        out = np.zeros(noutput_items, dtype=np.complex64)
        L = lib()
        L.grhip_fft_filter_ccc_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_fft_filter_ccc_work(self._h, int(noutput_items), _ptr(x), _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, d_out, stream=None):
        L = lib()
        L.grhip_fft_filter_ccc_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return _check(L.grhip_fft_filter_ccc_work_device(self._h, int(noutput_items), _devptr(d_in), _devptr(d_out),
                                                         _stream(stream)))


class pfb_channelizer_ccf(_Block):
    _destroy = "grhip_pfb_channelizer_ccf_destroy"

    def __init__(self, numchans, taps, oversample_rate=1, device=0):
        _Block.__init__(self)
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.numchans = int(numchans)
        L = lib()
        L.grhip_pfb_channelizer_ccf_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_void_p,
                                                       C.c_size_t, C.c_float, C.c_int]
        _check(L.grhip_pfb_channelizer_ccf_create(C.byref(self._h), self.numchans, _ptr(t), len(t),
                                                  float(oversample_rate), int(device)))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        L = lib()
        L.grhip_pfb_channelizer_ccf_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_pfb_channelizer_ccf_set_taps(self._h, _ptr(t), len(t)))

    def history(self):
        return _check(lib().grhip_pfb_channelizer_ccf_history(self._h))

    def output_multiple(self):
        return _check(lib().grhip_pfb_channelizer_ccf_output_multiple(self._h))

    def general_work(self, noutput_items, streams):
        """streams: list of numchans complex arrays each with history()-1 old
        items in front.  Returns (out[n, numchans], consumed)."""
        arrs = [np.ascontiguousarray(s, dtype=np.complex64) for s in streams]
        ptrs = (C.c_void_p * self.numchans)(*[a.ctypes.data for a in arrs])
        out = np.zeros((max(noutput_items, 1), self.numchans), dtype=np.complex64)
        consumed = C.c_int(0)
        L = lib()
        L.grhip_pfb_channelizer_ccf_general_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                             C.POINTER(C.c_int)]
        n = _check(L.grhip_pfb_channelizer_ccf_general_work(self._h, int(noutput_items), ptrs, _ptr(out),
                                                            C.byref(consumed)))
        return out[:n], consumed.value

    def general_work_device(self, noutput_items, d_in, stream_stride_items, d_out, stream=None):
        L = lib()
        L.grhip_pfb_channelizer_ccf_general_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p,
                                                                    C.c_size_t, C.c_void_p, C.c_void_p]
        return _check(L.grhip_pfb_channelizer_ccf_general_work_device(
            self._h, int(noutput_items), _devptr(d_in), int(stream_stride_items), _devptr(d_out),
            _stream(stream)))


def _pfb_hier_work_device(self, noutput_items, d_in, d_out, out_stride_items, stream=None):
    """blks2.pfb_channelizer_ccf (hier block) in one call: one interleaved stream in (taps_per_filter * numchans history
    items in front), numchans streams out (channel k at d_out + k * out_stride_items)"""
    L = lib()
    L.grhip_pfb_channelizer_ccf_hier_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                                             C.c_void_p]
    return _check(L.grhip_pfb_channelizer_ccf_hier_work_device(self._h, int(noutput_items), _devptr(d_in), _devptr(d_out),
                                                               int(out_stride_items), _stream(stream)))


pfb_channelizer_ccf.hier_work_device = _pfb_hier_work_device


class pfb_decimator_ccf(_Block):
    """gr.pfb_decimator_ccf(decim, taps, channel)"""
    _destroy = "grhip_pfb_decimator_ccf_destroy"

    def __init__(self, decim, taps, channel=0, device=0):
        _Block.__init__(self)
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.decim = int(decim)
        L = lib()
        L.grhip_pfb_decimator_ccf_create.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_void_p, C.c_size_t,
                                                     C.c_uint, C.c_int]
        _check(L.grhip_pfb_decimator_ccf_create(C.byref(self._h), self.decim, _ptr(t), len(t), int(channel),
                                                int(device)))

    def set_taps(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        L = lib()
        L.grhip_pfb_decimator_ccf_set_taps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.grhip_pfb_decimator_ccf_set_taps(self._h, _ptr(t), len(t)))

    def history(self):
        return _check(lib().grhip_pfb_decimator_ccf_history(self._h))

    def work(self, noutput_items, streams):
        arrs = [np.ascontiguousarray(s, dtype=np.complex64) for s in streams]
        ptrs = (C.c_void_p * self.decim)(*[a.ctypes.data for a in arrs])
        out = np.zeros(max(noutput_items, 1), dtype=np.complex64)
        L = lib()
        L.grhip_pfb_decimator_ccf_work.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = _check(L.grhip_pfb_decimator_ccf_work(self._h, int(noutput_items), ptrs, _ptr(out)))
        return out[:n]

    def work_device(self, noutput_items, d_in, stream_stride_items, d_out, stream=None):
        L = lib()
        L.grhip_pfb_decimator_ccf_work_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                                          C.c_void_p]
        return _check(L.grhip_pfb_decimator_ccf_work_device(self._h, int(noutput_items), _devptr(d_in),
                                                            int(stream_stride_items), _devptr(d_out), _stream(stream)))


# ----------------------------------------------------------------------------
# full DMR chain (multi-stream, device resident)
# ----------------------------------------------------------------------------
class _ChainParams(C.Structure):
    _fields_ = [("decimation", C.c_int), ("taps", C.c_void_p), ("ntaps", C.c_size_t),
                ("center_freq", C.c_double), ("sampling_freq", C.c_double), ("demod_gain", C.c_float),
                ("omega", C.c_float), ("gain_omega", C.c_float), ("mu", C.c_float), ("gain_mu", C.c_float),
                ("omega_relative_limit", C.c_float), ("access_code", C.c_char_p),
                ("access_code_len", C.c_size_t), ("threshold", C.c_int)]


class dmr_chain(_Block):
    _destroy = "grhip_dmr_chain_destroy"

    def __init__(self, decimation, taps, center_freq, sampling_freq, demod_gain, omega, gain_omega, mu,
                 gain_mu, omega_relative_limit, access_code, threshold, n_streams, max_samples, device=0):
        _Block.__init__(self)
        self._taps = np.ascontiguousarray(taps, dtype=np.complex64)
        code = access_code.encode("latin-1") if isinstance(access_code, str) else bytes(access_code)
        self._code = code
        p = _ChainParams(int(decimation), self._taps.ctypes.data, len(self._taps), float(center_freq),
                         float(sampling_freq), float(demod_gain), float(omega), float(gain_omega), float(mu),
                         float(gain_mu), float(omega_relative_limit), code, len(code), int(threshold))
        self.n_streams = int(n_streams)
        L = lib()
        L.grhip_dmr_chain_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(_ChainParams), C.c_int,
                                             C.c_size_t, C.c_int]
        _check(L.grhip_dmr_chain_create(C.byref(self._h), C.byref(p), self.n_streams, int(max_samples),
                                        int(device)))

    def set_mode(self, mode):
        _check(lib().grhip_dmr_chain_set_mode(self._h, int(mode)))

    def set_captures_per_wave(self, captures):
        L = lib()
        L.grhip_dmr_chain_set_captures_per_wave.argtypes = [C.c_void_p, C.c_int]
        _check(L.grhip_dmr_chain_set_captures_per_wave(self._h, int(captures)))

    def set_max_symbols(self, max_symbols):
        """noutput_items of the clock recovery per capture (0: unbounded)"""
        L = lib()
        L.grhip_dmr_chain_set_max_symbols.argtypes = [C.c_void_p, C.c_size_t]
        _check(L.grhip_dmr_chain_set_max_symbols(self._h, int(max_symbols)))

    def set_four_level(self, enable, pager_alpha=0.001):
        """4FSK tail: pager.slicer_fb(alpha) -> unpack_k_bits_bb(2) -> correlator; two output items per symbol"""
        L = lib()
        L.grhip_dmr_chain_set_four_level.argtypes = [C.c_void_p, C.c_int, C.c_float]
        _check(L.grhip_dmr_chain_set_four_level(self._h, int(bool(enable)), float(pager_alpha)))

    def run_device(self, d_in, n_samples, stream_stride_items, d_bits, bits_stride, d_nbits, stream=None):
        L = lib()
        L.grhip_dmr_chain_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                                 C.c_size_t, C.c_void_p, C.c_void_p]
        return _check(L.grhip_dmr_chain_run_device(self._h, _devptr(d_in), int(n_samples),
                                                   int(stream_stride_items), _devptr(d_bits),
                                                   int(bits_stride), _devptr(d_nbits), _stream(stream)))

    def intermediate(self, which):
        p = C.c_void_p(0)
        stride = C.c_size_t(0)
        L = lib()
        L.grhip_dmr_chain_intermediate.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_size_t)]
        _check(L.grhip_dmr_chain_intermediate(self._h, int(which), C.byref(p), C.byref(stride)))
        return p.value, stride.value


# ----------------------------------------------------------------------------
# minimal stand-in for the scheduler around ONE sync block / decimator:
# history-1 zeros preloaded (runtime/gr_flat_flowgraph.cc:150), then work() in
# chunks of at most `chunk` outputs (runtime/gr_block_executor.cc:76-78 caps a
# call at half a 64 KiB buffer), re-presenting the history in front of each call
# (runtime/gr_sync_decimator.cc:46-66).  A work() that returns 0 items (taps
# update) is simply called again, as the executor would.
# ----------------------------------------------------------------------------
def run_sync_block(block, x, chunk=4096, out_dtype=None):
    h = block.history()
    d = block.decimation()
    buf = np.concatenate([np.zeros(h - 1, dtype=x.dtype), x])
    n_total = len(x) // d
    outs = []
    done = 0
    retries = 0
    while done < n_total:
        n = min(chunk, n_total - done)
        seg = buf[done * d: done * d + n * d + h - 1]
        y = block.work(n, seg)
        if len(y) == 0:
            # history may have changed
            retries += 1
            if retries > 4:
                raise RuntimeError("block keeps returning 0 items")
            nh = block.history()
            if nh != h:
                # re-present with the new history length: keep alignment of the
                # newest item, like the scheduler's read pointer does
                x_pos = done * d
                raw = np.concatenate([np.zeros(nh - 1, dtype=x.dtype), x])
                buf = raw
                h = nh
                _ = x_pos
            continue
        outs.append(y)
        done += len(y)
    if not outs:
        return np.zeros(0, dtype=out_dtype or x.dtype)
    return np.concatenate(outs)
