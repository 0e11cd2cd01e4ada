"""Shared helpers for the test-suite: golden fixtures and record comparison."""
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
GROUPS = ("static", "md20", "synth64", "periodic8")

# Tolerance (relative) on window diameters.  north_star allows 1e-6; the kernels restate
# numpy's arccos (SVML), sin/cos and scalar ** (glibc) operation by operation, so against
# fixtures produced on the reference platform (glibc 2.35, AVX-512 numpy) EVERYTHING is
# bit-identical and the golden comparisons use 0.  Comparisons against the oracle run live
# use LIVE_TOL_WINDOW: 0 on that platform, north_star's 1e-6 elsewhere.
TOL_WINDOW = 0.0


def _reference_platform() -> bool:
    import platform

    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except ImportError:  # pragma: no cover
        return False
    libc = platform.libc_ver()
    return bool(feats.get("AVX512_SKX")) and libc[0] == "glibc" and libc[1] == "2.35"


LIVE_TOL_WINDOW = 0.0 if _reference_platform() else 1e-6


def load_group(tag):
    return np.load(GOLDEN / f"{tag}.npz")


def group_batch(g):
    """(atom_offset, xyz, vdw, mass) for a golden group."""
    from pywindow_amd import element_data as E

    ids = E.element_ids(g["elements"])
    return g["atom_offset"].astype(np.int64), np.ascontiguousarray(g["coordinates"]), E.VDW[ids], E.MASS[ids]


def molecules(g):
    off = g["atom_offset"]
    return [(g["elements"][off[u]:off[u + 1]], g["coordinates"][off[u]:off[u + 1]]) for u in range(len(off) - 1)]


def rel(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0


def check_records(recs, g, *, exact_scalars=True, tol_window=TOL_WINDOW, where=""):
    """Compare engine records (structured array / sequence of mappings) with a golden group."""
    n = len(g["atom_offset"]) - 1
    assert len(recs) == n
    stats = {"win_d": 0.0, "win_c_abs": 0.0}
    for u in range(n):
        r = recs[u]
        tag = f"{where} unit {u} ({g['names'][u]})"
        assert int(r["n_atoms"]) == int(g["n_atoms"][u]), tag
        for k in ("maxd_i", "maxd_j", "pore_atom", "pore_opt_atom", "n_windows"):
            assert int(r[k]) == int(g[k][u]), f"{tag}: {k} {r[k]} != {g[k][u]}"
        for k in ("mw", "maxd", "avg_d", "pore_d", "pore_vol", "pore_opt_d", "pore_vol_opt"):
            if exact_scalars:
                assert float(r[k]) == float(g[k][u]), f"{tag}: {k} {float(r[k])!r} != {float(g[k][u])!r}"
            else:
                assert rel(r[k], g[k][u]) <= 1e-12, f"{tag}: {k}"
        for k in ("com", "pore_opt_c"):
            if exact_scalars:
                assert np.array_equal(np.asarray(r[k]), g[k][u]), f"{tag}: {k}"
            else:
                assert rel(r[k], g[k][u]) <= 1e-12, f"{tag}: {k}"
        if "opt_nit" in getattr(r, "dtype", np.dtype([])).names or ():
            assert int(r["opt_nit"]) == int(g["st_opt_nit"][u]), f"{tag}: opt_nit"
            assert int(r["opt_nfev"]) == int(g["st_opt_nfev"][u]), f"{tag}: opt_nfev"
            assert int(r["n_points"]) == int(g["st_n_points"][u]), f"{tag}: n_points"
            npass = int(g["st_pass_offset"][u + 1] - g["st_pass_offset"][u])
            assert int(r["n_survivors"]) == npass, f"{tag}: n_survivors"
            if npass:
                assert float(r["eps"]) == float(g["st_eps"][u]), f"{tag}: eps"
        k = int(g["n_windows"][u])
        if k > 0 and tol_window == 0.0:
            # the ORDER of the windows is part of the result: ascending cluster label, which is what
            # iterating the reference's set of labels yields (utilities.py:1481-1523)
            assert np.array_equal(np.asarray(r["win_d"][:k]), g["win_d"][u][:k]), f"{tag}: window order / diameters"
            assert np.array_equal(np.asarray(r["win_c"][:k]).reshape(k, 3), g["win_c"][u][:k]), f"{tag}: window order / centres"
        if k > 0:
            # the reference's tests compare after sorting by diameter (tests/test_validate_cc3.py:423-439)
            p = np.argsort(np.asarray(r["win_d"][:k]))
            q = np.argsort(g["win_d"][u][:k])
            wd = np.asarray(r["win_d"][:k])[p]
            gd = g["win_d"][u][:k][q]
            e = rel(wd, gd)
            assert e <= tol_window, f"{tag}: window diameters rel err {e:.3e}"
            stats["win_d"] = max(stats["win_d"], e)
            wc = np.asarray(r["win_c"][:k]).reshape(k, 3)[p]
            gc = g["win_c"][u][:k][q]
            ea = float(np.max(np.abs(wc - gc)))
            assert ea <= (0.0 if tol_window == 0.0 else 1e-4), f"{tag}: window centres abs err {ea:.3e}"
            stats["win_c_abs"] = max(stats["win_c_abs"], ea)
    return stats


def check_stage_capture(dbg, g, where=""):
    """Stage-level parity: the capture of find_windows (``_lib.UNIT_DEBUG_DTYPE`` records, from the GPU
    through ``pw_analysis_debug`` or from the host build of the kernel source) against what the
    reference computed on the way (tests/golden/make_golden.py monkey-patches the reference's
    vector_analysis / DBSCAN / angle_between_vectors / minimize / brute): surviving sampling vectors,
    DBSCAN labels, and per window the chosen vector, the two rotation angles, the neck position,
    the z optimum, the in-plane optimum and the diameter.  Everything bit for bit."""
    n = len(g["atom_offset"]) - 1
    assert len(dbg) == n
    poff = g["st_pass_offset"]
    cols = {c: i for i, c in enumerate(g["win_table_cols"])}
    traced = {int(u) for u in g["trace_units"]} if "trace_units" in g.files else set()
    n_win = 0
    for u in range(n):
        tag = f"{where} unit {u} ({g['names'][u]})"
        d = dbg[u]
        ns = int(poff[u + 1] - poff[u])
        assert int(d["n_survivors"]) == ns, f"{tag}: survivors"
        assert np.array_equal(d["pass_idx"][:ns], g["st_pass_idx"][poff[u]:poff[u + 1]]), f"{tag}: pass_idx"
        assert np.array_equal(d["labels"][:ns], g["st_labels"][poff[u]:poff[u + 1]]), f"{tag}: DBSCAN labels"
        if u in traced and ns:
            res = g[f"tr{u}_pass_res"]                  # vector_analysis results [dist, 2m, p(3), v(3)]
            assert np.array_equal(d["gap2"][:ns], res[:, 1]), f"{tag}: path minima"
        rows = g["win_table"][g["win_unit"] == u]
        for c, row in enumerate(rows):
            if row[cols["ok"]] != 1.0:
                continue
            w = d["win"][c]
            assert np.array_equal(w[0:3], row[[cols["vx"], cols["vy"], cols["vz"]]]), f"{tag} window {c}: vector"
            assert w[3] == row[cols["angle_1"]] and w[4] == row[cols["angle_2"]], f"{tag} window {c}: angles"
            assert w[5] == -row[cols["z_lb"]], f"{tag} window {c}: neck position"
            assert w[7] == row[cols["z_x"]], f"{tag} window {c}: z optimum"
            assert w[8] == row[cols["xy_x"]] and w[9] == row[cols["xy_y"]], f"{tag} window {c}: in-plane optimum"
            assert w[10] == row[cols["diam"]], f"{tag} window {c}: diameter"
            n_win += 1
    return n_win
