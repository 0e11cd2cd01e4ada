"""Pins the ORACLE (oracle/pw_oracle.py + oracle/pw_prim.c) against golden vectors
produced by the reference itself (tests/golden/make_golden.py): bit-for-bit on
every quantity, optimiser outputs included."""
import numpy as np
import pytest

from _util import load_group, molecules

# (group, unit): the reference's own known-answer inputs + real MD + synthetic + periodic
CASES = [("static", 0), ("static", 1), ("static", 2), ("static", 6), ("md20", 0), ("md20", 12),
         ("synth64", 3), ("periodic8", 5)]


@pytest.mark.parametrize("tag,u", CASES)
def test_oracle_reproduces_reference_bit_for_bit(tag, u):
    from oracle import pw_oracle as O
    from pywindow_amd import element_data as E

    g = load_group(tag)
    el, xyz = molecules(g)[u]
    ids = E.element_ids(el)
    r = O.full_analysis(xyz, E.VDW[ids], E.MASS[ids])
    for k in ("mw", "maxd", "avg_d", "pore_d", "pore_vol", "pore_opt_d", "pore_vol_opt"):
        assert r[k] == float(g[k][u]), (k, r[k], float(g[k][u]))
    for k in ("maxd_i", "maxd_j", "pore_atom", "pore_opt_atom", "n_windows", "n_atoms"):
        assert r[k] == int(g[k][u]), k
    assert np.array_equal(r["com"], g["com"][u])
    assert np.array_equal(r["pore_opt_c"], g["pore_opt_c"][u])
    n = max(int(g["n_windows"][u]), 0)
    assert np.array_equal(r["win_d"][:n], g["win_d"][u][:n])
    assert np.array_equal(r["win_c"][:n], g["win_c"][u][:n])


def test_oracle_on_the_edge_tile_of_the_distance_matrix():
    """tests/golden/edge_tile.npz (written by the reference): five molecules whose farthest pair is an entry of the
    BLAS's edge tile (oracle/pw_prim.c: edge_order) -- every quantity, bit for bit."""
    import pathlib

    from oracle import pw_oracle as O
    from pywindow_amd import element_data as E

    g = np.load(pathlib.Path(__file__).resolve().parent / "golden" / "edge_tile.npz")
    for m in range(int(g["count"])):
        el, xyz = g["m%d_elements" % m], g["m%d_coordinates" % m]
        ids = E.element_ids(el)
        r = O.full_analysis(xyz, E.VDW[ids], E.MASS[ids])
        for k in ("maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert r[k] == float(g["m%d_%s" % (m, k)]), (m, k)
        assert (r["maxd_i"], r["maxd_j"]) == tuple(int(a) for a in g["m%d_maxd_atoms" % m]), m
        wd = g["m%d_win_d" % m]
        assert r["n_windows"] == len(wd), m
        assert np.array_equal(r["win_d"][:len(wd)], wd) and np.array_equal(r["win_c"][:len(wd)], g["m%d_win_c" % m]), m


def test_edge_tile_rule_against_sklearn_live():
    """The N x N restatement (oracle/pw_prim.c: pwo_max_dim with edge_order) against sklearn's own distance matrix on
    random molecules of every size class: below and above the BLAS's row panel (192), with and without an edge tile,
    up to the size from which the BLAS threads the product (383) with whatever thread count this machine gives, and
    beyond it with one BLAS thread.  The rule belongs to OpenBLAS's AVX-512 kernels."""
    import numpy  # noqa: F401  (loads the BLAS threadpoolctl reports on)
    threadpoolctl = pytest.importorskip("threadpoolctl")

    blas = [d for d in threadpoolctl.threadpool_info() if d.get("internal_api") == "openblas"]
    if not blas or any(d.get("architecture") != "SkylakeX" for d in blas):
        pytest.skip("the restated orders are those of OpenBLAS's SkylakeX kernels; this BLAS runs %r"
                    % [d.get("architecture") for d in blas])
    from sklearn.metrics.pairwise import euclidean_distances

    from oracle import pw_oracle as O

    rng = np.random.default_rng(12)
    radii = np.array([1.2, 1.7, 1.55, 1.52, 1.8])
    for n in [5, 12, 21, 45, 47, 78, 100, 119, 127, 172, 191, 196, 204, 255, 260, 316, 349, 380, 382] * 6:
        p = rng.normal(size=(n, 3))
        xyz = np.round(p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(3.0, 12.0) + rng.normal(scale=0.2, size=(n, 3)), 6)
        vdw = radii[rng.integers(0, int(rng.integers(1, 6)), size=n)]
        d = euclidean_distances(xyz, xyz) + (vdw[:, None] + vdw[None, :])       # utilities.py:366-370
        d = np.triu(d)
        i, j = np.unravel_index(np.argmax(d), d.shape)
        assert O.max_dim(O.Cage(xyz, vdw, np.ones(n))) == (int(i), int(j), float(d[i, j])), n
    # From 383 atoms OpenBLAS shares the product among its threads and the entries of the edge tile follow the thread
    # count: the platform restated there is ONE BLAS thread, whose level-3 panel recurrence (pw_prim.c:
    # last_panel_start) explains every entry probed up to 8197 atoms.  The deciding pair is forced onto an edge atom
    # (the four atoms of the last partial 8-block moved outwards), so that the edge rule decides the value.
    with threadpoolctl.threadpool_limits(limits=1, user_api="blas"):
        for n in [383, 388, 396, 412, 444, 476, 508, 572, 580, 700, 765, 772, 900, 1004, 1012, 1340] * 2:
            p = rng.normal(size=(n, 3))
            xyz = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(6.0, 12.0) + rng.normal(scale=0.2, size=(n, 3))
            if n % 8 >= 4:
                t0 = 8 * (n // 8)
                xyz[t0:t0 + 4] *= rng.uniform(1.05, 1.3)
            xyz = np.round(xyz, 6)
            vdw = radii[rng.integers(0, int(rng.integers(1, 6)), size=n)]
            d = np.triu(euclidean_distances(xyz, xyz) + (vdw[:, None] + vdw[None, :]))
            i, j = np.unravel_index(np.argmax(d), d.shape)
            assert O.max_dim(O.Cage(xyz, vdw, np.ones(n))) == (int(i), int(j), float(d[i, j])), n


def test_distance_primitive_matches_captured_objective_values():
    """The C primitive against objective values the reference evaluated through
    sklearn's euclidean_distances (L-BFGS-B evaluation traces in the fixtures)."""
    from oracle import pw_oracle as O
    from pywindow_amd import element_data as E

    for tag in ("static", "md20"):
        g = load_group(tag)
        mols = molecules(g)
        for u in g["trace_units"][:4]:
            el, xyz = mols[u]
            ids = E.element_ids(el)
            cage = O.Cage(xyz, E.VDW[ids], E.MASS[ids])
            tr = g[f"tr{u}_opt"]
            for row in tr[:: max(1, len(tr) // 60)]:
                assert -(cage.gap(row[:3])[0] * 2) == row[3]


def test_reference_literals():
    """Numbers printed in the reference's tests (tests/test_validate_cc3.py:357-412)."""
    from oracle import pw_oracle as O
    from pywindow_amd import element_data as E

    g = load_group("static")
    el, xyz = molecules(g)[0]
    ids = E.element_ids(el)
    cage = O.Cage(xyz, E.VDW[ids], E.MASS[ids])
    assert O.max_dim(cage) == (12, 54, 22.179369990077188)
    assert O.pore_diameter(cage) == (5.397020177310022, 29)
    assert O.molecular_weight(cage.mass) == 1117.5479999999998
    np.testing.assert_almost_equal(O.centre_of_mass(cage), [12.4, 12.4, 12.4])
