"""Non-default knobs of the reference's free functions:
``find_windows(adjust, pore_opt, increment)`` (utilities.py:1364-1371) and
``find_average_diameter(adjust)`` (:1586-1591).  Golden values come from the reference
itself (tests/golden/make_golden.py options)."""
import ctypes

import numpy as np
import pytest

from _util import GOLDEN, TOL_WINDOW, group_batch, rel
from pywindow_amd import _lib


def load_options():
    return np.load(GOLDEN / "options.npz")


def compare_windows(n, wd, wc, g, u, k, where):
    assert int(n) == int(g["n_windows"][u][k]), f"{where}: window count"
    m = int(n)
    if m <= 0:
        return 0.0
    p = np.argsort(np.asarray(wd[:m]))
    q = np.argsort(g["win_d"][u][k][:m])
    e = rel(np.asarray(wd[:m])[p], g["win_d"][u][k][:m][q])
    assert e <= TOL_WINDOW, f"{where}: window diameters rel err {e:.3e}"
    ea = np.max(np.abs(np.asarray(wc[:m]).reshape(m, 3)[p] - g["win_c"][u][k][:m][q]))
    assert ea == 0.0, f"{where}: window centres {ea:.3e}"
    return e


def test_oracle_options_match_reference():
    from oracle import pw_oracle as O

    g = load_options()
    off, xyz, vdw, mass = group_batch(g)
    for u in (0, 1):        # the oracle is slow: two molecules, every knob set
        cage = O.Cage(xyz[off[u]:off[u + 1]], vdw[off[u]:off[u + 1]], mass[off[u]:off[u + 1]])
        for k, (adjust, pore_opt, increment) in enumerate(g["window_options"]):
            res = O.find_windows(cage, adjust=adjust, pore_opt=bool(pore_opt), increment=increment)
            assert np.array_equal(np.sort(res[0]), np.sort(g["win_d"][u][k][: len(res[0])])), (u, k)
        for k, adjust in enumerate(g["average_options"]):
            assert O.find_average_diameter(cage, adjust) == g["avg_d"][u][k], (u, k)


def run_hostsim(hostsim, g, stages, prm):
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    off, xyz, vdw, mass = group_batch(g)
    vdw = np.ascontiguousarray(vdw)
    mass = np.ascontiguousarray(mass)
    out = np.zeros(len(off) - 1, dtype=_lib.UNIT_OUT_DTYPE)
    vp = ctypes.c_void_p
    rc = L.hs_analysis_batch(ctypes.c_long(len(off) - 1), off.ctypes.data_as(vp), xyz.ctypes.data_as(vp),
                             vdw.ctypes.data_as(vp), mass.ctypes.data_as(vp), ctypes.c_uint(stages),
                             out.ctypes.data_as(vp), ctypes.byref(prm))
    assert rc == 0
    return out


def test_host_team_options_match_reference(hostsim):
    g = load_options()
    for k, (adjust, pore_opt, increment) in enumerate(g["window_options"]):
        prm = _lib.Params(adjust_windows=adjust, pore_opt=bool(pore_opt), increment=increment)
        out = run_hostsim(hostsim, g, _lib.STAGE_WINDOWS, prm)
        for u in range(len(out)):
            compare_windows(out[u]["n_windows"], out[u]["win_d"], out[u]["win_c"], g, u, k, f"hostsim u{u} opt{k}")
    for k, adjust in enumerate(g["average_options"]):
        out = run_hostsim(hostsim, g, _lib.STAGE_AVG, _lib.Params(adjust_average=adjust))
        assert np.array_equal(out["avg_d"], g["avg_d"][:, k]), k


@pytest.mark.gpu
def test_hip_options_match_reference(hip_ctx):
    from pywindow_amd import utilities as U

    g = load_options()
    off = g["atom_offset"]
    worst = 0.0
    for u in range(len(off) - 1):
        el, xyz = g["elements"][off[u]:off[u + 1]], g["coordinates"][off[u]:off[u + 1]]
        for k, (adjust, pore_opt, increment) in enumerate(g["window_options"]):
            res = U.find_windows(el, xyz, adjust=adjust, pore_opt=bool(pore_opt), increment=increment)
            worst = max(worst, compare_windows(len(res[0]), res[0], res[1], g, u, k, f"hip u{u} opt{k}"))
        for k, adjust in enumerate(g["average_options"]):
            assert U.find_average_diameter(el, xyz, adjust=adjust) == g["avg_d"][u][k], (u, k)
        # defaults are restored after a call with knobs
        assert U.find_average_diameter(el, xyz) != g["avg_d"][u][1]
    print("options: worst window rel err", worst)


@pytest.mark.gpu
def test_params_validation(hip_ctx):
    with pytest.raises(_lib.PwHipError):
        hip_ctx.set_params(_lib.Params(increment=0.0))
    with pytest.raises(_lib.PwHipError):
        hip_ctx.set_params(_lib.Params(adjust_windows=-1.0))
    hip_ctx.set_params(None)
