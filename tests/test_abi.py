"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports
every function include/pywindow_amd.h declares, the record layout matches, and
compute calls FAIL LOUDLY without a HIP device (there is no CPU fallback)."""
import ctypes
import re

import numpy as np
import pytest

from _util import ROOT
from pywindow_amd import _lib


def header_functions():
    text = (ROOT / "include" / "pywindow_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pw_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = header_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(L, name), f"{name} declared in pywindow_amd.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_record_layout_matches_header(hostsim):
    hs = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    assert hs.hs_sizeof_unit_out() == _lib.UNIT_OUT_DTYPE.itemsize


def test_no_silent_cpu_fallback():
    L = _lib.load()
    if L.pw_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(_lib.PwHipError):
        _lib.Context(0)
    import pywindow_amd as pw

    mol = pw.Molecule({"elements": np.array(["C", "C"]), "coordinates": np.zeros((2, 3))}, "x", 0)
    with pytest.raises(_lib.PwHipError):
        mol.full_analysis()
    with pytest.raises(_lib.PwHipError):
        pw.pore_diameter(np.array(["C", "C"]), np.eye(3)[:2])


def test_unknown_element_raises_keyerror_like_reference():
    from pywindow_amd import engine

    with pytest.raises(KeyError):
        engine.make_batch([(np.array(["Zz"]), np.zeros((1, 3)))])


def test_product_package_never_imports_oracle():
    import pathlib

    for path in pathlib.Path(ROOT / "pywindow_amd").rglob("*.py"):
        src = path.read_text()
        assert "oracle" not in src.replace("the oracle", ""), path
        assert "hostsim" not in src, path
