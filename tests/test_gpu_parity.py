"""Parity tests proper: the HIP engine (through the C ABI) against the golden
vectors produced by the reference, against the oracle run live, and -- at the
benchmark's full size -- through size-independent properties."""
import numpy as np
import pytest

from _util import GOLDEN, GROUPS, LIVE_TOL_WINDOW, check_records, group_batch, load_group, molecules, rel

pytestmark = pytest.mark.gpu


def analyse_group(ctx, g, stages=15):
    from pywindow_amd import _lib

    off, xyz, vdw, mass = group_batch(g)
    return ctx.analyse(_lib.Batch(off, xyz, vdw, mass), stages)


@pytest.mark.parametrize("tag", GROUPS)
def test_hip_matches_reference_golden(hip_ctx, tag):
    g = load_group(tag)
    out = analyse_group(hip_ctx, g)
    stats = check_records(out, g, where=f"hip/{tag}")
    assert (out["status"] == 0).all()
    print(tag, "worst window rel err", stats)


def test_objective_is_bit_exact(hip_ctx):
    """min_i(|r_i - p| - vdw_i) at random points: bit-identical to the oracle's C
    primitive (itself bit-identical to sklearn's euclidean_distances)."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib

    g = load_group("static")
    off, xyz, vdw, mass = group_batch(g)
    batch = _lib.Batch(off, xyz, vdw, mass)
    rng = np.random.default_rng(7)
    n_units = len(off) - 1
    units = rng.integers(0, n_units, 4000)
    pts = np.empty((len(units), 3))
    cages = []
    for u in range(n_units):
        sl = slice(off[u], off[u + 1])
        cages.append(O.Cage(xyz[sl], vdw[sl], mass[sl]))
    for q, u in enumerate(units):
        pts[q] = cages[u].xyz.mean(0) + rng.normal(0, 3.0, 3)
    gap, arg = hip_ctx.point_gaps(batch, units, pts)
    for q, u in enumerate(units):
        v, i = cages[u].gap(pts[q])
        assert gap[q] == v and arg[q] == i, (q, u, gap[q], v)


def test_hip_matches_live_oracle(hip_ctx):
    """Same seeded inputs through the oracle (numpy/scipy/sklearn + C primitive) and the GPU."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(3, first=500)
    ids = E.element_ids(elements)
    out = hip_ctx.analyse(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
    for k in range(len(frames)):
        ref = O.full_analysis(frames[k], E.VDW[ids], E.MASS[ids])
        r = out[k]
        for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert float(r[key]) == ref[key], (k, key)
        assert np.array_equal(r["pore_opt_c"], ref["pore_opt_c"])
        assert int(r["n_windows"]) == ref["n_windows"]
        n = ref["n_windows"]
        assert rel(np.sort(r["win_d"][:n]), np.sort(ref["win_d"][:n])) <= LIVE_TOL_WINDOW


def test_stage_entry_points(hip_ctx):
    g = load_group("periodic8")
    basic = analyse_group(hip_ctx, g, stages=1)
    assert np.array_equal(basic["pore_d"], g["pore_d"]) and np.array_equal(basic["maxd"], g["maxd"])
    avg = analyse_group(hip_ctx, g, stages=2)
    assert np.array_equal(avg["avg_d"], g["avg_d"])
    opt = analyse_group(hip_ctx, g, stages=4)
    assert np.array_equal(opt["pore_opt_d"], g["pore_opt_d"])
    win = analyse_group(hip_ctx, g, stages=8)
    assert np.array_equal(win["n_windows"], g["n_windows"])


def test_full_size_properties(hip_ctx):
    """BASELINE config 2 (1000 CC3 frames): results do not depend on batch
    composition or order, and reproduce run to run."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(1000)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    res = hip_ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
    res.launch()
    a = res.download()
    res.launch()
    b = res.download()
    assert a.tobytes() == b.tobytes(), "not reproducible run to run"
    # the first 64 frames are the golden synth64 group
    g = load_group("synth64")
    check_records(a[:64], g, where="hip/1000-frame batch")
    # permuted order + a ragged neighbour (different N) in the same launch
    perm = np.random.default_rng(3).permutation(1000)[:200]
    sub = hip_ctx.analyse(_lib.Batch.uniform(frames[perm], vdw, mass))
    keys = [k for k in a.dtype.names]
    for k in keys:
        assert np.array_equal(sub[k], a[perm][k]), k
    assert (a["n_windows"] == 4).sum() > 950
    assert (a["status"] == 0).all()


def test_ragged_batch_and_edge_cases(hip_ctx):
    from pywindow_amd import _lib

    g = load_group("static")
    out_all = analyse_group(hip_ctx, g)
    mols = molecules(g)
    from pywindow_amd import engine

    # single-unit launches equal the batched launch
    for u in (1, 5, 10):
        one = engine.analyse([mols[u]])[0]
        for k in out_all.dtype.names:
            assert np.array_equal(one[k], out_all[u][k]), (u, k)
    # empty batch
    empty = hip_ctx.analyse(_lib.Batch(np.zeros(1, np.int64), np.zeros((0, 3)), np.zeros(0), np.zeros(0)))
    assert len(empty) == 0


def test_config3_periodic_cell_with_many_cages(hip_ctx):
    """BASELINE config 3: a periodic cell's worth of discrete cages (the 8 rebuilt cages of
    tests/data/system_periodic_rebuild.pdb, replicated 1x1x3 = 24 cages) in ONE launch; the
    replicas (pure translations by lattice vectors) must give translation-consistent results and
    the originals must match the reference."""
    from pywindow_amd import _lib

    g = load_group("periodic8")
    off, xyz, vdw, mass = group_batch(g)
    cell = 24.8
    reps = [xyz + np.array([0.0, 0.0, cell * k]) for k in range(3)]
    n8 = len(off) - 1
    off24 = np.concatenate([off[:-1] + k * off[-1] for k in range(3)] + [[3 * off[-1]]])
    out = hip_ctx.analyse(_lib.Batch(off24, np.concatenate(reps), np.tile(vdw, 3), np.tile(mass, 3)))
    check_records(out[:n8], g, where="config3/cell0")
    for k in (1, 2):
        rep = out[k * n8:(k + 1) * n8]
        assert np.array_equal(rep["n_windows"], g["n_windows"])
        # translated input: same geometry up to rounding of the shifted coordinates (atom
        # indices of near-degenerate extrema may legitimately differ on these symmetric cages)
        assert rel(rep["maxd"], g["maxd"]) < 1e-12 and rel(rep["pore_d"], g["pore_d"]) < 1e-12
        assert rel(rep["avg_d"], g["avg_d"]) < 1e-9
        assert np.max(np.abs(rep["com"] - (g["com"] + np.array([0, 0, cell * k])))) < 1e-9


def test_config5_screen_sample_against_live_oracle(hip_ctx):
    """BASELINE config 5 (perturbed cages x frames, throughput mode), sampled: seeds
    20260000 + 1000*cage + frame; a random sample is checked against the oracle run live."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, base = synth.load_cc3_base()
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    cages, frames = 40, 8
    coords = np.empty((cages * frames, len(base), 3))
    for c in range(cages):
        for f in range(frames):
            coords[c * frames + f] = synth.quantise_like_history(
                synth.noisy_frame(base, synth.SEED_BASE + 1000 * c + f, 0.10))
    out = hip_ctx.analyse(_lib.Batch.uniform(coords, vdw, mass))
    assert (out["status"] == 0).all()
    rng = np.random.default_rng(11)
    worst = 0.0
    for u in rng.choice(len(coords), 6, replace=False):
        ref = O.full_analysis(coords[u], vdw, mass)
        r = out[u]
        for key in ("maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert float(r[key]) == ref[key], (u, key)
        assert int(r["n_windows"]) == ref["n_windows"]
        n = ref["n_windows"]
        worst = max(worst, rel(np.sort(r["win_d"][:n]), np.sort(ref["win_d"][:n])))
    assert worst <= LIVE_TOL_WINDOW


def test_many_distinct_radii_on_the_gpu(hip_ctx):
    """More distinct radii than the kernels group by (ungrouped fall-back loops), against the oracle."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, base = synth.load_cc3_base()
    pool = ["C", "H", "N", "O", "S", "P", "F", "CL", "BR", "I", "SI", "SE", "ZN", "CU", "LI"]
    swapped = np.array([pool[i % len(pool)] if elements[i] == "H" else elements[i] for i in range(len(elements))])
    ids = E.element_ids(swapped)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    frames = np.array([synth.quantise_like_history(synth.noisy_frame(base, 777 + k, 0.05)) for k in range(3)])
    out = hip_ctx.analyse(_lib.Batch.uniform(frames, vdw, mass))
    for k in range(3):
        ref = O.full_analysis(frames[k], vdw, mass)
        r = out[k]
        for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert float(r[key]) == ref[key], (k, key)
        assert int(r["n_windows"]) == ref["n_windows"]
        n = max(ref["n_windows"], 0)
        assert rel(np.sort(r["win_d"][:n]), np.sort(ref["win_d"][:n])) <= LIVE_TOL_WINDOW


@pytest.mark.parametrize("copies", [3, 6])
def test_large_molecules_against_the_oracle(hip_ctx, copies):
    """Hundreds to a thousand atoms per unit (narrower teams, arrays spilling from LDS to the
    workspace): several CC3 cages side by side treated as one molecule, against the oracle."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, base = synth.load_cc3_base()
    ids1 = E.element_ids(elements)
    xyz = np.concatenate([synth.quantise_like_history(synth.noisy_frame(base, 4242 + k, 0.05)) + np.array([26.0 * k, 0.0, 0.0])
                          for k in range(copies)])
    vdw, mass = np.tile(E.VDW[ids1], copies), np.tile(E.MASS[ids1], copies)
    out = hip_ctx.analyse(_lib.Batch(np.array([0, len(xyz)], np.int64), xyz, vdw, mass))[0]
    ref = O.full_analysis(xyz, vdw, mass)
    for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
        assert float(out[key]) == ref[key], key
    assert (int(out["maxd_i"]), int(out["maxd_j"])) == (ref["maxd_i"], ref["maxd_j"])
    assert int(out["n_windows"]) == ref["n_windows"]
    n = max(ref["n_windows"], 0)
    assert rel(np.sort(out["win_d"][:n]), np.sort(ref["win_d"][:n])) <= LIVE_TOL_WINDOW


@pytest.mark.parametrize("tag", GROUPS)
def test_stage_capture_matches_reference(hip_ctx, tag):
    """Where a mismatch would come from: the intermediate results of find_windows read back from the
    GPU (pw_analysis_debug) against what the reference computed on the way -- surviving sampling
    vectors, DBSCAN labels, path minima, and per window the chosen vector, both rotation angles, the
    neck position, the z optimum, the in-plane optimum and the diameter, bit for bit."""
    from _util import check_stage_capture
    from pywindow_amd import _lib

    g = load_group(tag)
    off, xyz, vdw, mass = group_batch(g)
    out, dbg = hip_ctx.analyse_debug(_lib.Batch(off, xyz, vdw, mass))
    check_records(out, g, where=f"hip-debug/{tag}")
    n_win = check_stage_capture(dbg, g, where=f"hip/{tag}")
    assert n_win == int(np.maximum(g["n_windows"], 0).sum())
    # the capture does not disturb the analysis, and a later plain analysis is not captured
    plain = hip_ctx.analyse(_lib.Batch(off, xyz, vdw, mass))
    assert plain.tobytes() == out.tobytes()


def test_window_order_is_the_reference_order(hip_ctx):
    """Windows come out in ascending cluster label -- the order in which the reference's set of labels
    iterates (utilities.py:1481-1523) -- not merely as the same set: unsorted comparison."""
    for tag in GROUPS:
        g = load_group(tag)
        out = analyse_group(hip_ctx, g)
        for u in range(len(out)):
            k = int(g["n_windows"][u])
            if k > 0:
                assert np.array_equal(out["win_d"][u][:k], g["win_d"][u][:k]), (tag, u)
                assert np.array_equal(out["win_c"][u][:k], g["win_c"][u][:k]), (tag, u)


def _methane():
    g = np.load(GOLDEN / "nonporous.npz")       # inputs and the reference's behaviour (make_golden.py: run_nonporous)
    return g["elements"], g["coordinates"], g


def test_non_porous_molecule_fails_where_the_reference_fails(hip_ctx, monkeypatch):
    """A molecule whose centre of mass lies inside an atom: pore_diameter is negative (reference:
    (-3.4, 0)), the default box of opt_pore_diameter is inverted and SciPy raises ValueError from
    opt_pore_diameter / find_windows / full_analysis (values from the reference run in the development
    container).  The single-molecule API raises the same; batches flag the unit instead and go on --
    identically in the pipelined and the one-launch shape of the analysis."""
    import pywindow_amd as pw
    from pywindow_amd import _lib, engine

    el, xyz, ref = _methane()
    assert pw.pore_diameter(el, xyz) == (float(ref["pore_d"]), int(ref["pore_atom"])) == (-3.4, 0)
    assert len(set(ref["messages"])) == 1 and bool(ref["find_windows_no_pore_opt_is_none"])
    msg = str(ref["messages"][0])
    assert msg == engine.NEGATIVE_PORE_MESSAGE
    for call in (lambda: pw.opt_pore_diameter(el, xyz), lambda: pw.find_windows(el, xyz)):
        with pytest.raises(ValueError, match=msg):
            call()
    assert pw.find_windows(el, xyz, pore_opt=False) is None       # no optimisation, no bounds: the reference returns None
    mol = pw.MolecularSystem.load_system({"elements": el, "coordinates": xyz}).system_to_molecule()
    with pytest.raises(ValueError, match=msg):
        mol.full_analysis()
    # what the reference had filled in before SciPy stopped it
    assert list(mol.properties) == list(ref["property_keys"])
    assert mol.properties["maximum_diameter"] == {"diameter": float(ref["maxd"]), "atom_1": int(ref["maxd_atoms"][0]),
                                                  "atom_2": int(ref["maxd_atoms"][1])}
    assert mol.properties["average_diameter"] == float(ref["avg_d"])
    assert mol.properties["pore_diameter"] == {"diameter": float(ref["pore_d"]), "atom": int(ref["pore_atom"])}
    assert mol.properties["pore_volume"] == float(ref["pore_vol"])
    assert np.array_equal(mol.properties["centre_of_mass"], ref["com"])
    # in a batch: flagged, None windows, neighbours unaffected
    g = load_group("md20")
    cage = molecules(g)[2]
    recs = engine.analyse([cage, (el, xyz), cage])
    assert float(recs[1]["pore_d"]) == float(ref["pore_d"]) and float(recs[1]["avg_d"]) == float(ref["avg_d"])
    assert int(recs[1]["status"]) & _lib.ST_NEGATIVE_PORE and int(recs[1]["n_windows"]) == -1
    assert engine.windows_of(recs[1]) is None
    assert recs[0].tobytes() == recs[2].tobytes() and float(recs[0]["pore_opt_d"]) == g["pore_opt_d"][2]
    # one-launch shape of the same analysis (PW_FUSED=1): same record
    monkeypatch.setenv("PW_FUSED", "1")
    fused = _lib.Context(0)
    try:
        again = fused.analyse(engine.make_batch([cage, (el, xyz), cage]))
    finally:
        fused.close()
    assert again.tobytes() == recs.tobytes()


def test_interleaved_batches_of_different_shapes(hip_ctx):
    """Launches of different plans in flight on one context (another batch size, another stage mask)
    must not share team workspaces: interleaved asynchronous launches give the bytes of launches run
    one at a time."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(64)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    big = hip_ctx.upload(_lib.Batch.uniform(frames[:40], vdw, mass))
    small = hip_ctx.upload(_lib.Batch.uniform(frames[40:64], vdw, mass))
    WIN, ALL = _lib.STAGE_WINDOWS, _lib.STAGE_ALL
    plan = [(big, ALL), (small, WIN), (big, WIN), (small, ALL), (big, ALL), (small, ALL), (big, WIN), (small, WIN)]
    want = []
    for res, st in plan:                 # one at a time
        res.launch(st)
        want.append(res.download().tobytes())
    for rep in range(3):                 # all in flight, download only what is still current
        for i, (res, st) in enumerate(plan):
            res.launch(st)
            if i >= len(plan) - 2:
                pass
        got_small = small.download().tobytes()
        got_big = big.download().tobytes()
        assert got_small == want[7] and got_big == want[6], rep
    # pairwise: launch A then B without waiting, check both
    for (ra, sa), (rb_, sb) in zip(plan[:-1], plan[1:]):
        if ra is rb_:
            continue
        ra.launch(sa)
        rb_.launch(sb)
        a = ra.download().tobytes()
        b = rb_.download().tobytes()
        assert a == want[plan.index((ra, sa))] and b == want[plan.index((rb_, sb))]
    big.free()
    small.free()


def test_uniform_batch_constants_travel_once(hip_ctx):
    """``pw_batch_in.template_atoms``: one vdw / mass template for a batch of one molecule type gives the
    bytes of the per-atom form."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(12)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    tmpl = _lib.Batch.uniform(frames, vdw, mass)
    assert tmpl.template_atoms == 168 and len(tmpl.vdw) == 168
    off = np.arange(13, dtype=np.int64) * 168
    per_atom = _lib.Batch(off, frames.reshape(-1, 3), np.tile(vdw, 12), np.tile(mass, 12))
    assert hip_ctx.analyse(tmpl).tobytes() == hip_ctx.analyse(per_atom).tobytes()
    units = np.arange(12) % 12
    pts = frames.mean(axis=1)
    ga, ia = hip_ctx.point_gaps(tmpl, units, pts)
    gb, ib = hip_ctx.point_gaps(per_atom, units, pts)
    assert np.array_equal(ga, gb) and np.array_equal(ia, ib)
    sa, sb = hip_ctx.shape(tmpl), hip_ctx.shape(per_atom)
    assert sa.tobytes() == sb.tobytes()
    with pytest.raises(_lib.PwHipError):
        bad = _lib.Batch(np.array([0, 100, 268], np.int64), frames[:2].reshape(-1, 3)[:268], vdw, mass, template_atoms=168)
        hip_ctx.analyse(bad)
    # The two forms take different LOAD paths since round 6: a template's radius groups are worked out once on the host
    # (template_groups_build), a per-atom batch groups every unit on the device.  Molecules with one, three, eight and
    # ten distinct radii (more than eight: no grouping at all), radii in every order of first appearance.
    rng = np.random.default_rng(6)
    pool = np.unique(E.VDW[E.VDW > 0])
    for n_radii, n_atoms in ((1, 40), (3, 57), (8, 90), (10, 75), (4, 168)):
        radii = rng.permutation(pool)[:n_radii]
        v = radii[rng.integers(0, n_radii, size=n_atoms)]
        m = rng.uniform(1.0, 40.0, size=n_atoms)
        shell = rng.normal(size=(n_atoms, 3))
        shell = shell / np.linalg.norm(shell, axis=1)[:, None] * rng.uniform(7.0, 10.0)
        fr = shell[None] + rng.normal(0.0, 0.15, size=(5, n_atoms, 3))
        a = hip_ctx.analyse(_lib.Batch.uniform(fr, v, m))
        b = hip_ctx.analyse(_lib.Batch(np.arange(6, dtype=np.int64) * n_atoms, fr.reshape(-1, 3), np.tile(v, 5), np.tile(m, 5)))
        assert a.tobytes() == b.tobytes(), (n_radii, n_atoms)


def test_reference_platform_branch_is_reported():
    """Which tolerance the live-oracle comparisons use on this box is the FIRST thing a run prints (conftest.py:
    pytest_report_header -- 0 on the reference platform, glibc 2.35 + AVX-512 numpy; north_star's 1e-6 elsewhere),
    not a warning at its end; here only that the branch is one of the two."""
    import conftest

    line = conftest.pytest_report_header(None)[0]
    assert ("tolerance 0" in line) == (LIVE_TOL_WINDOW == 0.0) and LIVE_TOL_WINDOW in (0.0, 1e-6), line


def test_results_do_not_depend_on_what_the_cu_ran_before(hip_ctx):
    """Team-shared memory holds what the previous workgroup left.  A unit whose optimiser ends an
    iteration with every variable on a bound reads a row of the L-BFGS-B matrix WN1 that was never
    formed (SciPy reads the zeros of its fresh workspace there): found as one unit in 8192 whose
    result changed from run to run.  That unit, copied to every 4th position of a large random batch
    (persistent teams, arbitrary predecessors), must give one result -- the oracle's."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, base = synth.load_cc3_base()
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    units = 8192
    coords = base[None] + np.random.default_rng(99).normal(0.0, 0.10, size=(units,) + base.shape)
    special = coords[1577].copy()
    coords[0::4] = special
    ref = O.full_analysis(special, vdw, mass)
    res = hip_ctx.upload(_lib.Batch.uniform(coords, vdw, mass))
    first = None
    for _ in range(4):
        res.launch()
        out = res.download()
        copies = out[0::4]
        assert (copies["pore_opt_d"] == ref["pore_opt_d"]).all()
        assert (copies["opt_nit"] == copies["opt_nit"][0]).all()
        assert all(c.tobytes() == copies[0].tobytes() for c in copies[::37])
        if first is None:
            first = out.tobytes()
        assert out.tobytes() == first, "not reproducible run to run"
    res.free()


def test_other_cages_with_fresh_noise_against_live_oracle(hip_ctx):
    """Every molecule of the static golden group (60 to 468 atoms, 2 to 6 windows or none, different
    element sets) with noise no fixture has seen, in ONE ragged batch: GPU against the oracle run live
    (tests/tools/parity_variety.py is the larger version of this)."""
    from oracle import pw_oracle as O
    from pywindow_amd import _lib
    from pywindow_amd import element_data as E

    g = load_group("static")
    rng = np.random.default_rng(20261003)
    off, xyz, vdw, mass = group_batch(g)
    xyz = xyz + rng.normal(0.0, 0.05, size=xyz.shape)
    out = hip_ctx.analyse(_lib.Batch(off, xyz, vdw, mass))
    checked_windows = 0
    for u in range(len(off) - 1):
        sl = slice(off[u], off[u + 1])
        try:
            ref = O.full_analysis(xyz[sl], vdw[sl], mass[sl])
        except ValueError:                       # non-porous with this noise: the reference raises, the batch flags
            assert int(out[u]["status"]) & _lib.ST_NEGATIVE_PORE, u
            continue
        r = out[u]
        assert int(r["status"]) == 0, u
        for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
            assert float(r[key]) == ref[key], (u, key)
        assert (int(r["maxd_i"]), int(r["maxd_j"])) == (ref["maxd_i"], ref["maxd_j"]), u
        assert np.array_equal(r["pore_opt_c"], ref["pore_opt_c"]), u
        assert int(r["n_windows"]) == ref["n_windows"], u
        n = ref["n_windows"]
        if n > 0:
            assert rel(r["win_d"][:n], ref["win_d"][:n]) <= LIVE_TOL_WINDOW, u      # (window order included)
            checked_windows += n
    assert checked_windows >= 30


def test_every_launch_shape_gives_the_same_records(monkeypatch):
    """One analysis, two launch shapes, the same bytes: the default pipeline (optimiser chains | average diameter
    | window search, three launches) and PW_FUSED=1 (every stage in one team).  On the real MD frames and on the
    static molecules (60 to 468 atoms, none to six windows).  (Round 4's split window search -- a third and fourth
    shape -- was measured 4-25x slower and removed in round 5.)"""
    from pywindow_amd import _lib

    for tag in ("md20", "static"):
        g = load_group(tag)
        off, xyz, vdw, mass = group_batch(g)
        batch = _lib.Batch(off, xyz, vdw, mass)
        got = {}
        for name, env in (("pipeline", {}), ("one launch", {"PW_FUSED": "1"})):
            monkeypatch.delenv("PW_FUSED", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ctx = _lib.Context(0)
            assert ctx.pipelined == (name != "one launch")
            res = ctx.upload(batch)
            res.launch()
            got[name] = res.download()
            # ... and again on the same context (the second launch runs on another buffer set)
            res.launch()
            assert res.download().tobytes() == got[name].tobytes(), (tag, name)
            res.free()
            ctx.close()
        check_records(got["pipeline"], g, where=f"{tag} pipeline")
        assert got["one launch"].tobytes() == got["pipeline"].tobytes(), tag


def test_sampling_vectors_from_the_table_are_the_computed_ones(monkeypatch):
    """The context's table of unit vectors (three products a sampling vector) against the vectors computed per unit
    (PW_UNIT_TABLE=0: a sine and a cosine each), and against no tables at all (PW_NB_TABLES=0: the windowed neighbour
    search too): the same bytes, on the MD frames, the static molecules and the `adjust` fixtures' sphere sizes."""
    from pywindow_amd import _lib

    for tag in ("md20", "static"):
        g = load_group(tag)
        off, xyz, vdw, mass = group_batch(g)
        batch = _lib.Batch(off, xyz, vdw, mass)
        got = {}
        for name, env in (("table", {}), ("computed", {"PW_UNIT_TABLE": "0"}), ("no tables", {"PW_NB_TABLES": "0"})):
            monkeypatch.delenv("PW_UNIT_TABLE", raising=False)
            monkeypatch.delenv("PW_NB_TABLES", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ctx = _lib.Context(0)
            got[name] = ctx.analyse(batch, 15)
            ctx.close()
        check_records(got["table"], g, where=f"{tag} unit table")
        assert got["computed"].tobytes() == got["table"].tobytes(), tag
        assert got["no tables"].tobytes() == got["table"].tobytes(), tag


def test_device_equals_host_path_on_degenerate_molecules():
    """One, two, three (collinear), four atoms, duplicate atoms, a ring, a cube, a far-away molecule: whatever the host
    path (the same source for a one-lane team, pinned to the reference by the CPU suite) gives for them, the device gives
    too -- every field of every record, statuses included.  These inputs go through corners the CC3 frames never touch
    (teams with idle lanes, empty radius groups, candidate lists of one atom)."""
    from pywindow_amd import _lib, engine

    rng = np.random.default_rng(7)
    ring = [(np.cos(t) * 4.0, np.sin(t) * 4.0, 0.0) for t in np.linspace(0, 2 * np.pi, 12, endpoint=False)]
    cube = [(x, y, z) for x in (-3.0, 3.0) for y in (-3.0, 3.0) for z in (-3.0, 3.0)]
    shell = rng.normal(size=(60, 3))
    shell = 6.0 * shell / np.linalg.norm(shell, axis=1)[:, None]
    mols = [
        (["C"], [(0.0, 0.0, 0.0)]),
        (["C", "H"], [(0.0, 0.0, 0.0), (1.1, 0.0, 0.0)]),
        (["C", "C", "C"], [(0.0, 0.0, 0.0), (1.5, 0.0, 0.0), (3.0, 0.0, 0.0)]),
        (["C", "N", "O", "H"], [(0.0, 0.0, 0.0), (1.5, 0.0, 0.0), (0.0, 1.5, 0.0), (0.0, 0.0, 1.5)]),
        (["C", "C", "H", "H"], [(0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (2.0, 0.0, 0.0), (2.0, 0.0, 0.0)]),
        (["C"] * 12, ring),
        (["C", "N"] * 4, cube),
        (["C"] * 30 + ["H"] * 30, shell),
        (["C"] * 30 + ["H"] * 30, shell + 1.0e4),
        (["C", "H", "N", "O", "S", "P", "F", "Cl", "Br", "I"] * 6, shell),      # ten radii: more than the groups hold
    ]
    batch = [(np.array(el), np.array(xyz, dtype=np.float64)) for el, xyz in mols]
    host = engine.analyse(batch, stages=_lib.STAGE_ALL, device=-1)
    dev = engine.analyse(batch, stages=_lib.STAGE_ALL, device=0)
    for k in host.dtype.names:
        a, b = host[k], dev[k]
        same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
        assert same, (k, a, b)


def test_device_equals_host_path_on_random_molecules():
    """Sixty random molecules -- 5 to 400 atoms of up to seven elements on noisy shells, blobs and pairs of shells --
    through every stage on the device and on the host path: all fields of all records equal.  (Hollow, dense, tiny and
    lop-sided inputs: windows or none, any number of radius groups, optimisers that run into their bounds.)"""
    from pywindow_amd import _lib, engine

    rng = np.random.default_rng(20261004)
    pool = np.array(["C", "H", "N", "O", "S", "F", "Cl"])
    batch = []
    for k in range(60):
        n = int(rng.integers(5, 401))
        kind = k % 3
        p = rng.normal(size=(n, 3))
        if kind == 0:        # a noisy shell (a cage of sorts)
            p = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(3.0, 12.0) + rng.normal(scale=0.3, size=(n, 3))
        elif kind == 1:      # a blob
            p = p * rng.uniform(1.0, 6.0)
        else:                # two shells, one off centre
            r = np.where(rng.random(n) < 0.5, rng.uniform(4.0, 7.0), rng.uniform(9.0, 12.0))
            p = p / np.linalg.norm(p, axis=1)[:, None] * r[:, None]
            p[: n // 2] += rng.normal(scale=1.5, size=3)
        el = pool[rng.integers(0, int(rng.integers(1, len(pool) + 1)), size=n)]
        batch.append((el, p + rng.normal(scale=5.0, size=3)))
    host = engine.analyse(batch, stages=_lib.STAGE_ALL, device=-1)
    dev = engine.analyse(batch, stages=_lib.STAGE_ALL, device=0)
    bad = []
    for k in host.dtype.names:
        a, b = host[k], dev[k]
        same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
        if not same:
            rows = [u for u in range(len(host)) if not (np.array_equal(a[u], b[u], equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a[u], b[u]))]
            bad.append((k, rows[:5]))
    assert not bad, bad


def test_a_line_search_step_that_ends_on_a_bound_on_the_device(hip_ctx):
    """tests/test_host_context.py::test_a_line_search_step_that_ends_on_a_bound, on the device."""
    from pywindow_amd import _lib
    from pywindow_amd import element_data as E
    from test_host_context import check_bound_step

    g = np.load(GOLDEN / "bound_step.npz")
    el, xyz = g["elements"], g["coordinates"]
    ids = E.element_ids(el)
    rec = hip_ctx.analyse(_lib.Batch(np.array([0, len(xyz)], np.int64), xyz, E.VDW[ids], E.MASS[ids]), _lib.STAGE_ALL)[0]
    check_bound_step(rec)


def test_the_edge_tile_of_the_distance_matrix_on_the_device(hip_ctx):
    """tests/test_host_context.py::test_the_edge_tile_of_the_distance_matrix, on the device."""
    from pywindow_amd import _lib
    from test_host_context import check_edge_tile, edge_tile_batch

    g, batch = edge_tile_batch()
    check_edge_tile(g, hip_ctx.analyse(batch, _lib.STAGE_ALL))



def test_max_dim_beyond_the_blas_threading_size_on_the_device(hip_ctx):
    """tests/test_host_context.py::test_max_dim_beyond_the_blas_threading_size, on the device: 383 ... 1340 atoms, the
    deciding pair on the BLAS's edge tile, scikit-learn with one BLAS thread as the reference."""
    from test_host_context import max_dim_beyond_382

    tot, bad = max_dim_beyond_382(hip_ctx, seed=6)
    assert tot == 30 and bad == 0


def test_device_equals_host_path_on_large_molecules():
    """Molecules of 650 ... 1550 atoms: four wave teams whose window frames no longer fit LDS four at a time get fewer fit
    SLOTS (2, then 1: the windows of a unit are fitted in rounds -- plan_launch, round 5; rounds 1-4 ran such molecules on
    two- and one-wave teams).  Hollow shells with a few holes, so that windows exist; every field of every record
    against the host path, one molecule per analysis and all of them in one batch."""
    from pywindow_amd import _lib, engine

    rng = np.random.default_rng(20261005)
    pool = np.array(["C", "H", "N", "O"])
    batch = []
    for n in (650, 900, 1150, 1400, 1550):
        p = rng.normal(size=(n, 3))
        p = p / np.linalg.norm(p, axis=1)[:, None] * (5.0 + 0.004 * n)
        holes = rng.normal(size=(3, 3))
        holes /= np.linalg.norm(holes, axis=1)[:, None]
        keep = np.all((p / np.linalg.norm(p, axis=1)[:, None]) @ holes.T < 0.93, axis=1)
        p = p[keep] + rng.normal(scale=0.15, size=(int(keep.sum()), 3))
        batch.append((pool[rng.integers(0, 4, size=len(p))], p))
    host = engine.analyse(batch, stages=_lib.STAGE_ALL, device=-1)
    dev = engine.analyse(batch, stages=_lib.STAGE_ALL, device=0)
    assert (host["n_windows"] >= 1).sum() >= 3          # (the holes are found: the fit slots are exercised)
    for k in host.dtype.names:
        for u in range(len(batch)):
            same = np.array_equal(host[k][u], dev[k][u], equal_nan=True) if host[k].dtype.kind == "f" else np.array_equal(host[k][u], dev[k][u])
            assert same, (k, u, len(batch[u][0]))
    for u, mol in enumerate(batch):
        one = engine.analyse([mol], stages=_lib.STAGE_ALL, device=0)
        assert one[0].tobytes() == dev[u].tobytes(), u
