"""CPU tests of the boundary: the C-ABI library loads, exports every symbol
include/grhip.h declares, refuses to compute without a GPU (no CPU fallback),
and the product package never touches oracle/."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gnuradio-3.5.0-dmr_amd")


def _declared():
    src = open(os.path.join(ROOT, "include", "grhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(grhip_[a-z0-9_A-Z]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(g):
    lib = g.lib()
    names = _declared()
    assert len(names) > 60
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_exported_symbols_are_all_declared(g):
    out = subprocess.check_output(["nm", "-D", "--defined-only", g.lib_path()], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and "grhip_" in l.split()[-1]
                      and not l.split()[-1].startswith("_Z"))
    assert set(exported) == set(_declared())


def test_no_cpu_fallback_without_device(g):
    if g.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(g.GrhipError) as e:
        g.fir_filter_ccf(1, [1.0, 2.0])
    assert e.value.code == -5      # GRHIP_ENODEV
    assert "no CPU fallback" in str(e.value)


def test_strerror(g):
    assert g.strerror(0) == "ok"
    assert "range" in g.strerror(-2)


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".inc")) or f == "Makefile":
                s = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"pyoracle|liboracle|libgrref|import_oracle|oracle/", s) and f != "tables.inc":
                    bad.append(os.path.join(dirpath, f))
    # tables.inc only mentions the generator script in a comment
    assert not bad, bad


def test_workload_shapes(wl):
    x = wl.fsk4_capture(40_000, stream_id=5)
    assert x.dtype == np.complex64 and len(x) == 40_000
    assert np.array_equal(x, wl.fsk4_capture(40_000, stream_id=5))        # seeded
    assert not np.array_equal(x, wl.fsk4_capture(40_000, stream_id=6))
    t = wl.cfg2_proto_taps()
    assert len(t) == 256 and abs(t.real.sum() - 1.0) < 1e-5 and np.all(t.imag == 0)
    t64 = wl.lowpass_taps(64, 0.1, 1.0)
    assert len(t64) == 64 and abs(t64.sum() - 1.0) < 1e-5
    assert len(wl.access_code_string()) == 48
    # the carrier sits at -1.25 MHz
    X = np.abs(np.fft.fft(x[:32768]))
    f = np.fft.fftfreq(32768, 1 / wl.CFG2["fs"])
    assert abs(f[np.argmax(X)] - wl.CFG2["carrier"]) < 150e3
