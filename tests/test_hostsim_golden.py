"""The kernel SOURCE (pywindow_amd/csrc/*.hpp), compiled for the host with a
one-thread team, against every golden vector produced by the reference.  This
is how the control flow of the HIP kernels is checked in a container without a
GPU; the GPU build of the same headers is checked by test_gpu_parity.py."""
import ctypes

import numpy as np
import pytest

from _util import GROUPS, check_records, group_batch, load_group
from pywindow_amd import _lib


def run_hostsim(hostsim, g, stages=15):
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    off, xyz, vdw, mass = group_batch(g)
    vdw = np.ascontiguousarray(vdw)
    mass = np.ascontiguousarray(mass)
    out = np.zeros(len(off) - 1, dtype=_lib.UNIT_OUT_DTYPE)
    rc = L.hs_analysis_batch(ctypes.c_long(len(off) - 1), off.ctypes.data_as(ctypes.c_void_p),
                             xyz.ctypes.data_as(ctypes.c_void_p), vdw.ctypes.data_as(ctypes.c_void_p),
                             mass.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(stages),
                             out.ctypes.data_as(ctypes.c_void_p), None)
    assert rc == 0
    return out


def run_hostsim_debug(hostsim, g, stages=15):
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    assert L.hs_sizeof_unit_debug() == _lib.UNIT_DEBUG_DTYPE.itemsize
    off, xyz, vdw, mass = group_batch(g)
    vdw = np.ascontiguousarray(vdw)
    mass = np.ascontiguousarray(mass)
    out = np.zeros(len(off) - 1, dtype=_lib.UNIT_OUT_DTYPE)
    dbg = np.zeros(len(off) - 1, dtype=_lib.UNIT_DEBUG_DTYPE)
    rc = L.hs_analysis_debug(ctypes.c_long(len(off) - 1), off.ctypes.data_as(ctypes.c_void_p),
                             xyz.ctypes.data_as(ctypes.c_void_p), vdw.ctypes.data_as(ctypes.c_void_p),
                             mass.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint(stages),
                             out.ctypes.data_as(ctypes.c_void_p), dbg.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out, dbg


@pytest.mark.parametrize("tag", ["static", "md20", "periodic8"])
def test_host_team_stage_capture_matches_reference(hostsim, tag):
    """Intermediate results of find_windows (survivors, DBSCAN labels, per-window angles / neck /
    optima) of the kernel source against the reference's, bit for bit."""
    from _util import check_stage_capture

    g = load_group(tag)
    out, dbg = run_hostsim_debug(hostsim, g)
    check_records(out, g, where=tag)
    assert check_stage_capture(dbg, g, where=tag) == int(np.maximum(g["n_windows"], 0).sum())


@pytest.mark.parametrize("tag", GROUPS)
def test_host_team_matches_reference(hostsim, tag):
    g = load_group(tag)
    out = run_hostsim(hostsim, g)
    stats = check_records(out, g, where=tag)
    # window diameters and centres are bit-identical too
    assert stats["win_d"] == 0.0 and stats["win_c_abs"] == 0.0
    assert (out["status"] == 0).all()


def test_reference_known_answers(hostsim):
    """Literals of the reference's own tests (tests/test_validate_cc3.py:353-439,
    test_validate_windows.py:1944-2087, test_validate_average_diameter.py:2373-2415)."""
    g = load_group("static")
    out = run_hostsim(hostsim, g)
    names = list(g["names"])
    cc3 = out[names.index("cc3")]
    np.testing.assert_almost_equal(cc3["com"], [12.4, 12.4, 12.4])
    assert cc3["maxd"] == 22.179369990077188 and (cc3["maxd_i"], cc3["maxd_j"]) == (12, 54)
    np.testing.assert_almost_equal(cc3["avg_d"], 13.832017514255472, decimal=7)
    assert cc3["pore_d"] == 5.397020177310022
    assert cc3["pore_vol"] == 82.31154385154417
    assert cc3["pore_opt_d"] == 5.397020177310022
    np.testing.assert_almost_equal(np.sort(cc3["win_d"][:4]),
                                   np.sort([3.63778746, 3.63562103, 3.62896512, 3.63707237]), decimal=7)
    assert out[names.index("windows_case_1")]["n_windows"] == -1   # C60: no windows -> None
    for case, nwin in (("windows_case_2", 2), ("windows_case_3", 3), ("windows_case_4", 4), ("windows_case_5", 6)):
        assert out[names.index(case)]["n_windows"] == nwin
    np.testing.assert_almost_equal(np.sort(out[names.index("windows_case_2")]["win_d"][:2]),
                                   np.sort([3.72937988, 3.34146021]), decimal=3)
    for case, avg in (("avgdiam_case_1", 12.38895620), ("avgdiam_case_2", 13.36606775), ("avgdiam_case_3", 18.10740925),
                      ("avgdiam_case_4", 19.23547068), ("avgdiam_case_5", 24.03139233)):
        np.testing.assert_almost_equal(out[names.index(case)]["avg_d"], avg, decimal=3)


def test_stage_selection(hostsim):
    g = load_group("periodic8")
    basic = run_hostsim(hostsim, g, stages=1)
    assert (basic["n_windows"] == -1).all() and (basic["avg_d"] == 0).all()
    assert np.array_equal(basic["pore_d"], g["pore_d"])
    opt = run_hostsim(hostsim, g, stages=4)
    assert np.array_equal(opt["pore_opt_d"], g["pore_opt_d"])


def test_many_distinct_radii_fall_back_to_ungrouped_loops(hostsim):
    """More distinct van der Waals radii than the kernels group by: the ungrouped code paths
    (per-atom square roots) must give the oracle's results too."""
    from oracle import pw_oracle as O
    from pywindow_amd import element_data as E
    from pywindow_amd import synth

    elements, base = synth.load_cc3_base()
    pool = ["C", "H", "N", "O", "S", "P", "F", "CL", "BR", "I", "SI", "SE", "ZN", "CU", "LI"]
    assert len({E.atomic_vdw_radius[e] for e in pool}) > 8
    keep = np.array([e for e in elements])
    swapped = np.array([pool[i % len(pool)] if keep[i] == "H" else keep[i] for i in range(len(keep))])
    ids = E.element_ids(swapped)
    vdw, mass = np.ascontiguousarray(E.VDW[ids]), np.ascontiguousarray(E.MASS[ids])
    g = {"atom_offset": np.array([0, len(base)], np.int64)}
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    out = np.zeros(1, dtype=_lib.UNIT_OUT_DTYPE)
    xyz = np.ascontiguousarray(base)
    vp = ctypes.c_void_p
    rc = L.hs_analysis_batch(ctypes.c_long(1), g["atom_offset"].ctypes.data_as(vp), xyz.ctypes.data_as(vp),
                             vdw.ctypes.data_as(vp), mass.ctypes.data_as(vp), ctypes.c_uint(15),
                             out.ctypes.data_as(vp), None)
    assert rc == 0
    ref = O.full_analysis(xyz, vdw, mass)
    r = out[0]
    for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
        assert float(r[key]) == ref[key], key
    assert (int(r["maxd_i"]), int(r["maxd_j"])) == (ref["maxd_i"], ref["maxd_j"])
    assert int(r["n_windows"]) == ref["n_windows"]
    n = max(ref["n_windows"], 0)
    assert np.array_equal(np.sort(r["win_d"][:n]), np.sort(ref["win_d"][:n]))
