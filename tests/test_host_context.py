"""``device = -1``: the explicit host path of the C ABI (BASELINE.json configs[0] "CC3 single-frame
full_analysis() on CPU").  ``pw_context_create(-1)`` runs the kernel SOURCE (pywindow_amd/csrc/pw_unit.hpp)
compiled by g++ for a one-lane team on host threads -- through the product's own boundary, so these tests
read like the GPU ones: every golden unit produced by the reference, bit for bit.  Nothing ever selects
this path implicitly (tests/test_abi.py::test_no_silent_cpu_fallback)."""
import ctypes

import numpy as np
import pytest

from _util import GOLDEN, GROUPS, check_records, group_batch, load_group, molecules
from pywindow_amd import _lib, engine
from pywindow_amd import element_data as E


@pytest.fixture(scope="module")
def host_ctx():
    return _lib.Context(-1, host_threads=4)


@pytest.mark.parametrize("tag", GROUPS)
def test_host_context_matches_reference(host_ctx, tag):
    g = load_group(tag)
    off, xyz, vdw, mass = group_batch(g)
    out = host_ctx.analyse(_lib.Batch(off, xyz, vdw, mass))
    stats = check_records(out, g, where=f"host {tag}")
    assert stats["win_d"] == 0.0 and stats["win_c_abs"] == 0.0
    assert (out["status"] == 0).all()


def test_host_context_threads_do_not_change_results():
    g = load_group("synth64")
    off, xyz, vdw, mass = group_batch(g)
    one = _lib.Context(-1, host_threads=1).analyse(_lib.Batch(off, xyz, vdw, mass))
    many = _lib.Context(-1, host_threads=7).analyse(_lib.Batch(off, xyz, vdw, mass))
    assert one.tobytes() == many.tobytes()


def test_config0_cc3_full_analysis_on_cpu():
    """BASELINE.json configs[0]: the reference's known-answer CC3 input (tests/test_validate_cc3.py:353-439)
    through Molecule.full_analysis() with the host context selected explicitly."""
    import pywindow_amd as pw

    g = load_group("static")
    el, xyz = molecules(g)[list(g["names"]).index("cc3")]
    engine.set_default_device(-1)
    try:
        props = pw.Molecule({"elements": el, "coordinates": xyz}, "cc3", 0).full_analysis()
    finally:
        engine.set_default_device(None)
    np.testing.assert_almost_equal(props["centre_of_mass"], [12.4, 12.4, 12.4])
    assert props["maximum_diameter"] == {"diameter": 22.179369990077188, "atom_1": 12, "atom_2": 54}
    assert props["pore_diameter"]["diameter"] == 5.397020177310022
    assert props["pore_volume"] == 82.31154385154417
    assert props["pore_diameter_opt"]["diameter"] == 5.397020177310022
    np.testing.assert_almost_equal(props["average_diameter"], 13.832017514255472, decimal=7)
    np.testing.assert_almost_equal(np.sort(props["windows"]["diameters"]),
                                   np.sort([3.63778746, 3.63562103, 3.62896512, 3.63707237]), decimal=7)


def test_host_context_has_no_cliffs(host_ctx):
    """The capacity fixtures (tests/test_cliffs.py) through the ABI: the host context sizes its workspace
    from the adjust knob, grows it when a unit asks, and delivers windows beyond PW_W_MAX."""
    from test_cliffs import check_window_call, load_cliffs, params_of

    mols, calls = load_cliffs()
    for call in calls:
        el, xyz = mols[call["mol"]]
        ids = E.element_ids(el)
        prm, stages = params_of(call)
        extra = []
        rec = host_ctx.analyse(_lib.Batch(np.array([0, len(xyz)], np.int64), xyz, E.VDW[ids], E.MASS[ids]), stages, prm, extra)[0]
        if call["kind"] == "avg":
            assert float(rec["avg_d"]) == call["avg_d"], call["label"]
        else:
            check_window_call(call, rec, extra[0] if extra else np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE))


def check_bound_step(rec):
    """A record against tests/golden/bound_step.npz (written by the reference, tests/golden/make_bound_step.py)."""
    g = np.load(GOLDEN / "bound_step.npz")
    assert int(rec["status"]) == 0
    for k in ("maxd", "avg_d", "pore_d", "pore_opt_d"):
        assert float(rec[k]) == float(g[k]), k
    assert np.array_equal(rec["pore_opt_c"], g["pore_opt_c"])
    assert (int(rec["opt_nit"]), int(rec["opt_nfev"])) == (int(g["opt_nit"]), int(g["opt_nfev"]))
    nw = len(g["win_d"])
    assert int(rec["n_windows"]) == nw
    assert np.array_equal(rec["win_d"][:nw], g["win_d"]) and np.array_equal(np.asarray(rec["win_c"]).reshape(-1)[:3 * nw], g["win_c"].reshape(-1))


def test_a_line_search_step_that_ends_on_a_bound(host_ctx):
    """SciPy's L-BFGS-B puts an iterate of the line search back on a bound it overshot by rounding (lnsrlb: "take step
    and prevent rounding error beyond bound").  One random molecule in 574 takes such a step -- stp == stpmx -- and
    without the projection ended two ulps outside its box, twelve objective evaluations and 1e-11 A away from the
    reference.  The fixture is that molecule with the reference's own results."""
    g = np.load(GOLDEN / "bound_step.npz")
    el, xyz = g["elements"], g["coordinates"]
    ids = E.element_ids(el)
    rec = host_ctx.analyse(_lib.Batch(np.array([0, len(xyz)], np.int64), xyz, E.VDW[ids], E.MASS[ids]), _lib.STAGE_ALL)[0]
    check_bound_step(rec)


def edge_tile_batch():
    """The molecules of tests/golden/edge_tile.npz as one batch, and the fixture."""
    g = np.load(GOLDEN / "edge_tile.npz")
    mols = [(g["m%d_elements" % m], g["m%d_coordinates" % m]) for m in range(int(g["count"]))]
    ids = [E.element_ids(el) for el, _ in mols]
    offsets = np.concatenate([[0], np.cumsum([len(x) for _, x in mols])]).astype(np.int64)
    batch = _lib.Batch(offsets, np.concatenate([x for _, x in mols]), np.concatenate([E.VDW[i] for i in ids]),
                       np.concatenate([E.MASS[i] for i in ids]))
    return g, batch


def check_edge_tile(g, recs):
    """Records against tests/golden/edge_tile.npz (written by the reference, tests/golden/make_edge_tile.py)."""
    for m, rec in enumerate(recs):
        assert int(rec["status"]) == 0, m
        for k in ("maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert float(rec[k]) == float(g["m%d_%s" % (m, k)]), (m, k)
        assert (int(rec["maxd_i"]), int(rec["maxd_j"])) == tuple(int(a) for a in g["m%d_maxd_atoms" % m]), m
        assert np.array_equal(rec["pore_opt_c"], g["m%d_pore_opt_c" % m]), m
        wd, wc = g["m%d_win_d" % m], g["m%d_win_c" % m]
        assert int(rec["n_windows"]) == len(wd), m
        assert np.array_equal(rec["win_d"][:len(wd)], wd), m
        assert np.array_equal(np.asarray(rec["win_c"]).reshape(-1)[:3 * len(wd)], wc.reshape(-1)), m


def test_the_edge_tile_of_the_distance_matrix(host_ctx):
    """sklearn's N x N distance matrix goes through OpenBLAS's dsyrk, whose kernel for the four atoms
    [8*(N//8), 8*(N//8)+4) of a molecule with N % 8 >= 4 sums the three products of an entry in another order than the
    body of the matrix does, by column chunk (pw_unit.hpp: GramEdgeRule).  Five random molecules in 1700 have their
    farthest pair -- of the molecule, or of the molecule shifted to its centre of mass, the radius of the sampling
    spheres -- on such an entry and came out one ulp away; the fixture is those five with the reference's results."""
    g, batch = edge_tile_batch()
    check_edge_tile(g, host_ctx.analyse(batch, _lib.STAGE_ALL))


def test_a_path_that_would_never_end_is_an_error(host_ctx):
    """find_windows walks radius / increment points along every sampling vector (and radius / increment2 along a
    cluster's refined one); the reference builds those paths as Python lists.  A pore centre that an open search box let
    run away -- or, here, a step of a millionth of an angstrom -- means millions of points per path: the unit gets
    PW_ST_PATH_TOO_LONG and no windows instead of a launch that runs for hours, the single-molecule call raises
    MemoryError, and the other units of the batch are analysed as ever."""
    import pywindow_amd as pw
    from pywindow_amd import engine

    g = load_group("static")
    el, xyz = molecules(g)[0]
    ids = E.element_ids(el)
    off = np.array([0, len(xyz), 2 * len(xyz)], np.int64)
    batch = _lib.Batch(off, np.concatenate([xyz, xyz]), np.tile(E.VDW[ids], 2), np.tile(E.MASS[ids], 2))
    ok = host_ctx.analyse(batch, _lib.STAGE_ALL)
    rec = host_ctx.analyse(batch, _lib.STAGE_ALL, _lib.Params(increment2=1.0e-6))
    assert (rec["status"] & _lib.ST_PATH_TOO_LONG).all() and (rec["n_windows"] == -1).all()
    for k in ("maxd", "avg_d", "pore_d", "pore_opt_d", "pore_opt_c"):          # everything but the windows is there
        assert np.array_equal(rec[k], ok[k]), k
    with pytest.raises(MemoryError):
        engine.raise_on_capacity(rec[0])
    with pytest.raises(MemoryError):
        engine.raise_on_uncomputable(rec)
    fine = host_ctx.analyse(batch, _lib.STAGE_ALL, _lib.Params(increment2=0.05))
    assert (fine["status"] == 0).all() and (fine["n_windows"] == 4).all()
    engine.set_default_device(-1)
    try:
        with pytest.raises(MemoryError):
            pw.utilities.find_windows(el, xyz, increment2=1.0e-6)
    finally:
        engine.set_default_device(None)


def test_host_context_resident_and_trajectory(tmp_path, host_ctx):
    """The trajectory driver on the host context: the reference's own 20-frame DL_POLY file, frames 3..6,
    against the md20 golden group."""
    from pywindow_amd.trajectory import DLPOLY

    g = np.load(load_group.__globals__["GOLDEN"] / "history20.npz")
    path = tmp_path / "HISTORY_singlemol_short"
    path.write_bytes(g["file_bytes"].tobytes())
    traj = DLPOLY(path)
    recs = traj.analysis_records(frames=[3, 4, 5, 6], swap_atoms={"he": "H"}, forcefield="opls", device=-1)
    md = load_group("md20")
    for k, f in enumerate((3, 4, 5, 6)):
        assert float(recs[k]["pore_opt_d"]) == float(md["pore_opt_d"][f])
        n = int(md["n_windows"][f])
        assert int(recs[k]["n_windows"]) == n and np.array_equal(recs[k]["win_d"][:n], md["win_d"][f][:n])
    # a batch kept "resident" (host memory here)
    off, xyz, vdw, mass = group_batch(md)
    res = host_ctx.upload(_lib.Batch(off[:4], xyz[: off[3]], vdw[: off[3]], mass[: off[3]]))
    res.launch()
    assert np.array_equal(res.download()["pore_opt_d"], md["pore_opt_d"][:3])
    res.free()


def test_host_context_serves_the_analysis_only(host_ctx):
    with pytest.raises(_lib.PwHipError, match="host path"):
        host_ctx.dbscan(np.zeros((8, 3)), 1.0)
    with pytest.raises(_lib.PwHipError, match="host path"):
        host_ctx.pairwise_sum(np.ones(8))
    assert not host_ctx.pipelined
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.pw_context_create(-2, ctypes.byref(h)) == -1        # PW_E_NO_DEVICE: only -1 names the host


def max_dim_beyond_382(ctx, seed=5, rounds=2):
    """max_dim of molecules of 383 ... 1340 atoms whose deciding pair has an atom of the BLAS's edge tile, against
    scikit-learn's distance matrix computed with ONE BLAS thread (from 383 atoms OpenBLAS threads the product and the
    reference's last bit follows the thread count; the platform restated is one thread -- DESIGN.md section 7,
    utilities.py:355-372).  Returns (molecules, mismatches)."""
    threadpoolctl = pytest.importorskip("threadpoolctl")
    blas = [d for d in threadpoolctl.threadpool_info() if d.get("internal_api") == "openblas"]
    if not blas or any(d.get("architecture") != "SkylakeX" for d in blas):
        pytest.skip("the restated orders are those of OpenBLAS's SkylakeX kernels")
    from sklearn.metrics.pairwise import euclidean_distances

    rng = np.random.default_rng(seed)
    radii = np.array([1.2, 1.7, 1.55, 1.52, 1.8])
    tot = bad = 0
    with threadpoolctl.threadpool_limits(limits=1, user_api="blas"):
        for n in [383, 388, 396, 412, 444, 476, 508, 572, 580, 700, 765, 772, 900, 1004, 1340] * rounds:
            p = rng.normal(size=(n, 3))
            xyz = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(6.0, 12.0) + rng.normal(scale=0.2, size=(n, 3))
            if n % 8 >= 4:
                t0 = 8 * (n // 8)
                xyz[t0:t0 + 4] *= rng.uniform(1.05, 1.3)
            xyz = np.round(xyz, 6)
            vdw = radii[rng.integers(0, int(rng.integers(1, 6)), size=n)]
            d = np.triu(euclidean_distances(xyz, xyz) + (vdw[:, None] + vdw[None, :]))
            i, j = np.unravel_index(np.argmax(d), d.shape)
            r = ctx.analyse(_lib.Batch(np.array([0, n], np.int64), xyz, vdw, np.ones(n)), _lib.STAGE_BASIC)[0]
            tot += 1
            bad += not (int(r["maxd_i"]) == int(i) and int(r["maxd_j"]) == int(j) and float(r["maxd"]) == float(d[i, j]))
    return tot, bad


def test_max_dim_beyond_the_blas_threading_size(host_ctx):
    tot, bad = max_dim_beyond_382(host_ctx)
    assert tot == 30 and bad == 0
