"""INTEGRATION.md's ctypes binding is real code: it is extracted from the document and executed.
Without a HIP device it must fail loudly THROUGH pw_last_error() (a function, called as one); on a
GPU box the patched ``full_analysis`` reproduces a golden unit."""
import re
import sys
import types

import numpy as np
import pytest

from _util import ROOT, load_group, molecules
from pywindow_amd import _lib


def snippet():
    text = (ROOT / "INTEGRATION.md").read_text()
    part = text[text.index("## 2."):]
    code = re.search(r"```python\n(.*?)```", part, flags=re.S).group(1)
    assert "pw_last_error()" in code and "in_dll" not in code
    return code.replace('"libpywindow_hip.so"', repr(str(_lib.LIB_PATH)))


@pytest.fixture()
def reference_tables_shim(monkeypatch):
    """``pywindow._internal.tables`` as the snippet imports it, served from the product's own tables
    (identical to the reference's: tests/test_history_and_driver.py::test_element_tables_equal_the_reference)."""
    from pywindow_amd import element_data as E

    pkg, internal, tables = types.ModuleType("pywindow"), types.ModuleType("pywindow._internal"), types.ModuleType("pywindow._internal.tables")
    tables.atomic_mass, tables.atomic_vdw_radius = E.atomic_mass, E.atomic_vdw_radius
    pkg._internal, internal.tables = internal, tables
    for name, mod in (("pywindow", pkg), ("pywindow._internal", internal), ("pywindow._internal.tables", tables)):
        monkeypatch.setitem(sys.modules, name, mod)


def test_snippet_fails_loudly_without_a_device(reference_tables_shim):
    if _lib.load().pw_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        exec(compile(snippet(), "INTEGRATION.md", "exec"), {})


@pytest.mark.gpu
def test_snippet_reproduces_a_golden_unit(reference_tables_shim):
    ns = {}
    exec(compile(snippet(), "INTEGRATION.md", "exec"), ns)
    g = load_group("md20")
    el, xyz = molecules(g)[5]

    class Mol:
        pass

    mol = Mol()
    mol.no_of_atoms, mol.elements, mol.coordinates, mol.properties = len(el), el, xyz, {"no_of_atoms": len(el)}
    props = ns["full_analysis"](mol)
    assert props["pore_diameter_opt"]["diameter"] == g["pore_opt_d"][5]
    assert np.array_equal(props["windows"]["diameters"], g["win_d"][5][: int(g["n_windows"][5])])
