"""Capacities the reference does not have: the number of sampling vectors
``int(log10(4 pi R^2) * 250 * adjust)`` (utilities.py:1409, 1616), the number of windows (:1481-1536)
and the molecule size are unbounded there, so here a limit of the engine may never show in a result.
``tests/golden/cliffs.npz`` holds what the REFERENCE returns (values, exceptions, logged warnings) for
large / tiny ``adjust``, for constructed molecules with more windows than a record holds, and for
constructed dropped / negative windows (tests/golden/make_golden.py cliffs)."""
import ctypes
import json
import logging

import numpy as np
import pytest

from _util import GOLDEN
from pywindow_amd import _lib
from pywindow_amd import element_data as E


def load_cliffs():
    g = np.load(GOLDEN / "cliffs.npz")
    mols = {}
    off = g["atom_offset"]
    for k, name in enumerate(g["mol_names"]):
        mols[str(name)] = (g["elements"][off[k]:off[k + 1]], np.ascontiguousarray(g["coordinates"][off[k]:off[k + 1]]))
    calls = []
    for i, label in enumerate(g["labels"]):
        calls.append({"label": str(label), "kind": str(g["kinds"][i]), "mol": str(g["mol_names"][g["mol_of_call"][i]]),
                      "kwargs": json.loads(str(g["kwargs"][i])), "n_windows": int(g["n_windows"][i]),
                      "win_d": g["win_d"][i], "win_c": g["win_c"][i], "avg_d": float(g["avg_d"][i]),
                      "centre_ndim": int(g["centre_ndim"][i]), "error": str(g["error"][i]),
                      "dropped": int(g["dropped"][i]), "negative": int(g["negative"][i])})
    return mols, calls


def params_of(call):
    kw = call["kwargs"]
    if call["kind"] == "avg":
        return _lib.Params(adjust_average=kw.get("adjust", 1)), _lib.STAGE_AVG
    return _lib.Params(adjust_windows=kw.get("adjust", 1), increment=kw.get("increment", 1.0),
                       increment2=kw.get("increment2", 0.1), pore_opt=kw.get("pore_opt", True)), _lib.STAGE_WINDOWS


def hostsim_run(hostsim, el, xyz, stages, prm, p_cap=0):
    """One molecule through the kernel source compiled for the host; (record, extra windows)."""
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    ids = E.element_ids(el)
    vdw, mass = np.ascontiguousarray(E.VDW[ids]), np.ascontiguousarray(E.MASS[ids])
    off = np.array([0, len(xyz)], np.int64)
    out = np.zeros(1, dtype=_lib.UNIT_OUT_DTYPE)
    xw = np.zeros(256, dtype=_lib.EXTRA_WINDOW_DTYPE)
    cnt = ctypes.c_uint(0)
    vp = ctypes.c_void_p
    rc = L.hs_analysis_ext(ctypes.c_long(1), off.ctypes.data_as(vp), xyz.ctypes.data_as(vp), vdw.ctypes.data_as(vp),
                           mass.ctypes.data_as(vp), ctypes.c_uint(stages), out.ctypes.data_as(vp), ctypes.byref(prm),
                           ctypes.c_int(p_cap), xw.ctypes.data_as(vp), ctypes.c_uint(len(xw)), ctypes.byref(cnt))
    assert rc == 0
    assert cnt.value <= len(xw)
    return out[0], xw[: cnt.value]


def check_window_call(call, rec, xw):
    lbl = call["label"]
    st = int(rec["status"])
    if call["error"]:
        assert "ValueError" in call["error"] and st & _lib.ST_TOO_FEW_POINTS, lbl
        return 0
    assert not st & (_lib.ST_POINTS_OVERFLOW | _lib.ST_TOO_FEW_POINTS), f"{lbl}: status {st}"
    n = call["n_windows"]
    assert int(rec["n_windows"]) == n, f"{lbl}: {int(rec['n_windows'])} windows, the reference {n}"
    assert bool(st & _lib.ST_WINDOW_DROPPED) == (call["dropped"] > 0), lbl
    assert bool(st & _lib.ST_WINDOW_NEGATIVE) == (call["negative"] > 0), lbl
    if n <= 0:
        return 0
    k = min(n, _lib.W_MAX)
    d = np.concatenate([rec["win_d"][:k], xw["d"]])
    c = np.concatenate([rec["win_c"][:k], xw["c"].reshape(-1, 3)])
    assert len(d) == n, f"{lbl}: {len(d)} of {n} windows delivered"
    if n > k:
        assert st & _lib.ST_WINDOW_OVERFLOW and list(xw["index"]) == list(range(k, n)), lbl
    # bit for bit, in the reference's order
    assert np.array_equal(d, call["win_d"][:n]), f"{lbl}: diameters"
    assert np.array_equal(c, call["win_c"][:n]), f"{lbl}: centres"
    return n


def test_fixture_covers_the_cliffs():
    mols, calls = load_cliffs()
    kinds = {(c["kind"], c["kwargs"].get("adjust")) for c in calls}
    for a in (2.6, 3.0, 4.0, 0.05):
        assert ("win", a) in kinds
    for a in (2.2, 3.0):
        assert ("avg", a) in kinds
    assert any(c["n_windows"] > _lib.W_MAX for c in calls)                 # more windows than a record holds
    assert any(c["error"] for c in calls)                                  # fewer than ten sampling vectors
    assert any(c["n_windows"] == 0 and c["centre_ndim"] == 1 for c in calls)   # all noise: two empty arrays
    assert any(c["dropped"] for c in calls) and any(c["negative"] for c in calls)   # the two logged warnings


def test_host_team_has_no_cliffs(hostsim):
    mols, calls = load_cliffs()
    delivered = 0
    for call in calls:
        el, xyz = mols[call["mol"]]
        prm, stages = params_of(call)
        rec, xw = hostsim_run(hostsim, el, xyz, stages, prm)
        if call["kind"] == "avg":
            assert float(rec["avg_d"]) == call["avg_d"], call["label"]
        else:
            delivered += check_window_call(call, rec, xw)
    assert delivered > 0


def test_too_small_workspace_is_flagged_never_a_value(hostsim):
    """A launch whose workspace is smaller than a unit asks for (only possible when the capacity is
    forced below what the adjust knob implies) flags the unit: NaN / None, and how many it wanted."""
    mols, calls = load_cliffs()
    el, xyz = mols["cc3"]
    rec, _ = hostsim_run(hostsim, el, xyz, _lib.STAGE_AVG, _lib.Params(adjust_average=3.0), p_cap=2048)
    assert int(rec["status"]) & _lib.ST_POINTS_OVERFLOW and np.isnan(rec["avg_d"]) and int(rec["n_points_avg"]) > 2048
    rec, _ = hostsim_run(hostsim, el, xyz, _lib.STAGE_WINDOWS, _lib.Params(adjust_windows=3.0), p_cap=2048)
    assert int(rec["status"]) & _lib.ST_POINTS_OVERFLOW and int(rec["n_windows"]) == -1 and int(rec["n_points"]) > 2048


def test_windows_of_shapes():
    """None / two empty arrays (centres of shape (0,), utilities.py:1526-1536) / arrays."""
    from pywindow_amd import engine

    rec = np.zeros(1, dtype=_lib.UNIT_OUT_DTYPE)[0]
    rec["n_windows"] = -1
    assert engine.windows_of(rec) is None
    rec["n_windows"] = 0
    d, c = engine.windows_of(rec)
    assert d.shape == (0,) and c.shape == (0,)
    rec["n_windows"] = 18
    rec["status"] = _lib.ST_WINDOW_OVERFLOW
    with pytest.raises(_lib.PwHipError):
        engine.windows_of(rec)
    more = (np.arange(2.0), np.ones((2, 3)))
    d, c = engine.windows_of(rec, more)
    assert d.shape == (18,) and c.shape == (18, 3)
    props = engine.records_to_properties(np.array([rec]), extra=np.array(
        [(0, 16, 0, 5.0, (1, 2, 3)), (0, 17, 0, 6.0, (4, 5, 6))], dtype=_lib.EXTRA_WINDOW_DTYPE))
    assert props[0]["windows"]["diameters"].shape == (18,) and props[0]["windows"]["diameters"][17] == 6.0


@pytest.mark.gpu
def test_hip_has_no_cliffs(hip_ctx, caplog):
    """The same fixture through the C ABI on the GPU: pw_analysis_batch sizes the workspace from the
    adjust knob, delivers the windows beyond PW_W_MAX through the extra-window list, and the façade
    returns / raises / logs what the reference does."""
    from pywindow_amd import utilities as U

    mols, calls = load_cliffs()
    largest = 0
    for call in calls:
        el, xyz = mols[call["mol"]]
        prm, stages = params_of(call)
        extra = []
        batch = _lib.Batch(np.array([0, len(xyz)], np.int64), xyz, E.VDW[E.element_ids(el)], E.MASS[E.element_ids(el)])
        rec = hip_ctx.analyse(batch, stages, prm, extra)[0]
        largest = max(largest, hip_ctx.point_capacity)
        if call["kind"] == "avg":
            assert float(rec["avg_d"]) == call["avg_d"], call["label"]
            assert U.find_average_diameter(el, xyz, **call["kwargs"]) == call["avg_d"]
            continue
        xw = extra[0] if extra else np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE)
        check_window_call(call, rec, xw)
        # the reference-shaped call
        caplog.clear()
        if call["error"]:
            with pytest.raises(ValueError):
                U.find_windows(el, xyz, **call["kwargs"])
            continue
        with caplog.at_level(logging.WARNING, logger="pywindow_amd"):
            res = U.find_windows(el, xyz, **call["kwargs"])
        n = call["n_windows"]
        if n < 0:
            assert res is None
        else:
            assert np.array_equal(res[0], call["win_d"][:n])
            assert np.array(res[1]).ndim == call["centre_ndim"]
            if n:
                assert np.array_equal(res[1], call["win_c"][:n])
        assert any("returned as None" in r.getMessage() for r in caplog.records) == (call["dropped"] > 0), call["label"]
        assert any("smaller than 0" in r.getMessage() for r in caplog.records) == (call["negative"] > 0), call["label"]
    assert largest >= 24000                        # adjust = 12 went through (25 000 sampling vectors per molecule) ...
    hip_ctx.analyse(_lib.Batch(np.array([0, len(mols["cc3"][1])], np.int64), mols["cc3"][1],
                               E.VDW[E.element_ids(mols["cc3"][0])], E.MASS[E.element_ids(mols["cc3"][0])]))
    assert hip_ctx.point_capacity < 8500           # ... and the workspace is small again once the knobs are


@pytest.mark.gpu
def test_hip_resident_path_delivers_extra_windows(hip_ctx):
    """A batch kept on the device (the trajectory drivers' route): the first download allocates the
    extra-window list and repeats the launch by itself."""
    mols, calls = load_cliffs()
    call = next(c for c in calls if c["n_windows"] > _lib.W_MAX)
    el, xyz = mols[call["mol"]]
    cc3 = mols["cc3"]
    ids, ids3 = E.element_ids(el), E.element_ids(cc3[0])
    off = np.array([0, len(cc3[1]), len(cc3[1]) + len(xyz), 2 * len(cc3[1]) + len(xyz)], np.int64)
    batch = _lib.Batch(off, np.concatenate([cc3[1], xyz, cc3[1]]), np.concatenate([E.VDW[ids3], E.VDW[ids], E.VDW[ids3]]),
                       np.concatenate([E.MASS[ids3], E.MASS[ids], E.MASS[ids3]]))
    res = hip_ctx.upload(batch)
    try:
        for _ in range(2):
            res.launch(_lib.STAGE_ALL)
            extra = []
            recs = res.download(extra)
            assert [int(r["n_windows"]) for r in recs] == [4, call["n_windows"], 4]
            xw = extra[0]
            assert (xw["unit"] == 1).all()
            check_window_call(call, recs[1], xw)
    finally:
        res.free()


def _two_atoms(distance):
    """Two atoms `distance` apart (they set the size of the sampling sphere) and a third one near the middle
    for the rays to meet."""
    xyz = np.zeros((3, 3))
    xyz[1, 0] = distance
    xyz[2] = (distance / 2.0, 5.0, 0.0)
    return _lib.Batch(np.array([0, 3], np.int64), xyz, np.full(3, 1.70), np.full(3, 12.011))


def test_host_context_grows_the_workspace_when_a_unit_asks():
    """Two atoms 12 000 A apart: the average-diameter sphere wants 2300 rays, more than the 2176
    the default knobs imply.  pw_analysis_batch raises the capacity and repeats; the value equals a run that
    had the capacity from the start."""
    ctx = _lib.Context(-1, host_threads=2)
    before = ctx.point_capacity
    rec = ctx.analyse(_two_atoms(12000.0), _lib.STAGE_AVG)[0]
    assert int(rec["n_points_avg"]) > before and not int(rec["status"]) & _lib.ST_POINTS_OVERFLOW
    assert np.isfinite(rec["avg_d"]) and rec["avg_d"] > 0
    roomy = _lib.Context(-1, host_threads=1).analyse(_two_atoms(12000.0), _lib.STAGE_AVG, _lib.Params(adjust_average=1.0))[0]
    assert float(roomy["avg_d"]) == float(rec["avg_d"])


def test_host_context_extra_windows_with_threads():
    """Several molecules with more than 16 windows in one batch, analysed by four threads: every unit gets
    its own extra windows, in order, whatever the threads did."""
    from pywindow_amd import engine

    mols, calls = load_cliffs()
    call = next(c for c in calls if c["n_windows"] > _lib.W_MAX)
    el, xyz = mols[call["mol"]]
    cc3 = mols["cc3"]
    batch = engine.make_batch([(el, xyz), cc3, (el, xyz), (el, xyz), cc3, (el, xyz)])
    extra = []
    recs = _lib.Context(-1, host_threads=4).analyse(batch, _lib.STAGE_WINDOWS, None, extra)
    more = engine.extra_by_unit(extra)
    assert sorted(more) == [0, 2, 3, 5]
    for u in (0, 2, 3, 5):
        d, c = engine.windows_of(recs[u], more[u])
        assert np.array_equal(d, call["win_d"][: call["n_windows"]]) and np.array_equal(c, call["win_c"][: call["n_windows"]])
    assert int(recs[1]["n_windows"]) == 4 and int(recs[4]["n_windows"]) == 4


@pytest.mark.gpu
def test_hip_grows_the_workspace_when_a_unit_asks(hip_ctx):
    """The same on the GPU, against the host context; and the resident path, which cannot repeat by itself,
    flags the unit (NaN, never a value) and the wrappers raise."""
    from pywindow_amd import engine

    want = _lib.Context(-1, host_threads=1).analyse(_two_atoms(12000.0), _lib.STAGE_AVG)[0]
    fresh = _lib.Context(0)
    try:
        assert fresh.point_capacity < int(want["n_points_avg"])
        res = fresh.upload(_two_atoms(12000.0))
        res.launch(_lib.STAGE_AVG)
        rec = res.download()[0]
        res.free()
        assert int(rec["status"]) & _lib.ST_POINTS_OVERFLOW and np.isnan(rec["avg_d"])
        with pytest.raises(_lib.PwHipError):
            engine.raise_on_capacity(rec)
        got = fresh.analyse(_two_atoms(12000.0), _lib.STAGE_AVG)[0]
        assert float(got["avg_d"]) == float(want["avg_d"]) and int(got["n_points_avg"]) == int(want["n_points_avg"])
        assert fresh.point_capacity >= int(want["n_points_avg"])
    finally:
        fresh.close()


@pytest.mark.gpu
def test_hip_large_and_small_molecules_in_one_batch(hip_ctx):
    """A batch whose largest molecule does not fit LDS runs from global memory as a whole: the small
    molecules beside it still come out as the reference has them."""
    from pywindow_amd import engine

    mols, calls = load_cliffs()
    big = mols["shell_big"]
    call = next(c for c in calls if c["mol"] == "shell_big" and c["kind"] == "win")
    avg = next(c for c in calls if c["mol"] == "shell_big" and c["kind"] == "avg")
    cc3_win = next(c for c in calls if c["mol"] == "windows_case_5" and c["kwargs"].get("adjust") == 3.0)
    del cc3_win
    from _util import load_group, molecules

    g = load_group("static")
    static = molecules(g)
    batch = engine.make_batch([static[0], big, static[5]])
    recs = hip_ctx.analyse(batch)
    assert int(recs[1]["n_atoms"]) == len(big[0]) > 1700
    assert int(recs[1]["n_windows"]) == call["n_windows"]
    assert np.array_equal(recs[1]["win_d"][: call["n_windows"]], call["win_d"][: call["n_windows"]])
    assert float(recs[1]["avg_d"]) == avg["avg_d"]
    for u, k in ((0, 0), (2, 5)):
        assert float(recs[u]["pore_opt_d"]) == float(g["pore_opt_d"][k]) and float(recs[u]["avg_d"]) == float(g["avg_d"][k])
        n = int(g["n_windows"][k])
        assert int(recs[u]["n_windows"]) == n and np.array_equal(recs[u]["win_d"][: max(n, 0)], g["win_d"][k][: max(n, 0)])


