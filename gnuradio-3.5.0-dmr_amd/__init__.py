"""grhip -- Python face of libgrhip.so (ctypes over the C ABI in include/grhip.h).

Mirrors the names the reference's SWIG layer gives the same blocks
(gr.fir_filter_ccf, gr.freq_xlating_fir_filter_ccc, gr.quadrature_demod_cf,
digital.clock_recovery_mm_ff, digital.correlate_access_code_bb,
digital.binary_slicer_fb, pager.slicer_fb, gr.unpack_k_bits_bb, gr.fft_vcc, gr.fft_filter_ccc,
gr.pfb_channelizer_ccf) with the same
constructor arguments; `work()` takes/returns numpy arrays with the
gr_sync_block contract (history items in front of the input).

There is no CPU fallback: every call goes to the HIP library and raises if it
is missing or no GPU is present.  The directory name is not a Python
identifier; load it with `import_grhip()` from grhip_loader.py (repo root).
"""
from .binding import *  # noqa: F401,F403
from .binding import __all__ as _b_all
from . import workload  # noqa: F401

__all__ = list(_b_all) + ["workload"]
