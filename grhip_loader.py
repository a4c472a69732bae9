"""Import the package directory `gnuradio-3.5.0-dmr_amd/` (not a valid Python
identifier) under the module name `grhip`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "gnuradio-3.5.0-dmr_amd")


def import_grhip():
    if "grhip" in sys.modules:
        return sys.modules["grhip"]
    spec = importlib.util.spec_from_file_location(
        "grhip", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["grhip"] = mod
    spec.loader.exec_module(mod)
    return mod


def import_oracle():
    """TEST-ONLY: the CPU checker (oracle/pyoracle.py).  Never imported by the
    product package."""
    p = os.path.join(_ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import pyoracle
    return pyoracle
