"""Periodic pre-processing: create_supercell + discrete_molecules
(reference utilities.py:768-1085).  Golden outputs were produced by the reference itself
(tests/golden/make_golden.py rebuild): CC3 cell as in tests/test_molecular.py:4467-4553 (33
fragments / 8 rebuilt cages), the same cell centred on the origin, two noisy MD-like
frames, a hexagonal and a triclinic cell, a framework that fills the supercell, and
non-periodic inputs."""
import ctypes

import numpy as np
import pytest

from _util import GOLDEN
from pywindow_amd import rebuild as RB


def load_cases():
    g = np.load(GOLDEN / "rebuild.npz")
    cases = {}
    for name in g["names"]:
        name = str(name)
        system = {k.split("__in_")[1]: g[k] for k in g.files if k.startswith(f"{name}__in_")}
        expect = {k.split("__")[1]: g[k] for k in g.files if k.startswith(f"{name}__") and "__in_" not in k}
        cases[name] = (system, expect)
    return cases


CASES = load_cases()
pytestmark = pytest.mark.filterwarnings("ignore::PendingDeprecationWarning")
SMALL = ["cc3_cell", "cc3_cell_centred", "cc3_cell_md0", "EPIRUR", "TATVER", "cc3_molecule", "saygor", "pudxes_xyz"]


def check_molecules(mols, expect, prefix, where):
    off = expect[f"{prefix}_offset"]
    assert len(mols) == len(off) - 1, f"{where}: {len(mols)} molecules, reference {len(off) - 1}"
    for k, m in enumerate(mols):
        lo, hi = off[k], off[k + 1]
        assert np.array_equal(m["elements"], expect[f"{prefix}_elements"][lo:hi]), f"{where}: mol {k} elements"
        assert np.array_equal(m["coordinates"], expect[f"{prefix}_xyz"][lo:hi]), f"{where}: mol {k} coordinates"
        if "atom_ids" in m:
            assert np.array_equal(m["atom_ids"], expect[f"{prefix}_ids"][lo:hi]), f"{where}: mol {k} atom ids"


@pytest.mark.parametrize("name", SMALL + ["MIBQAR"])
def test_oracle_matches_reference(name):
    from oracle import pw_rebuild as R

    system, expect = CASES[name]
    check_molecules(R.discrete_molecules(system), expect, "plain", f"oracle/{name}")
    if "rebuild_offset" in expect:
        check_molecules(R.discrete_molecules(system, rebuild=R.create_supercell(system)), expect, "rebuild",
                        f"oracle/{name}/rebuild")


def test_oracle_primitives():
    """The two arithmetic restatements the supercell rests on: numpy's 3x3 matrix-vector
    product and Python's round(x, 8)."""
    from oracle import pw_rebuild as R

    rng = np.random.default_rng(5)
    m = rng.normal(size=(3, 3)) * 10
    pts = rng.normal(size=(400, 3)) * 20
    ref = np.array([np.array(np.matrix(m) * p.reshape(-1, 1)).reshape(-1) for p in pts])
    assert np.array_equal(R.mat3_apply(m, pts), ref)
    k = rng.integers(-5e9, 5e9, 20000)
    for x in (rng.normal(size=20000) * 30, (k + 0.5) / 1e8, np.nextafter((k + 0.5) / 1e8, np.inf), k / 1e8):
        assert np.array_equal(R.round8(x), np.array([round(float(v), 8) for v in x]))


def run_hostsim(hostsim, system, rebuild, with_bits=True):
    L = ctypes.CDLL(str(hostsim / "librebuildprobe.so"))
    topo = RB.CellTopology(system["elements"])
    lattice, periodic = RB.system_lattice(system)
    if rebuild and lattice is None:
        lattice = np.asarray(system["lattice"], float)
    xyz, lat, inv = RB.pack_frames(np.asarray(system["coordinates"], float)[None], None if lattice is None else lattice[None])
    n = topo.n
    cap = 30 * n if rebuild else n
    n_mol = ctypes.c_int()
    status = ctypes.c_int()
    off = np.zeros(cap + 1, np.int32)
    src = np.zeros(cap, np.int32)
    img = np.zeros(cap, np.int8)
    oxyz = np.zeros((cap, 3))
    vp = ctypes.c_void_p
    rc = L.hs_discrete_molecules(
        ctypes.c_int(n), xyz.ctypes.data_as(vp), None if lat is None else lat.ctypes.data_as(vp),
        None if inv is None else inv.ctypes.data_as(vp), topo.cov.ctypes.data_as(vp), topo.mass.ctypes.data_as(vp),
        topo.terminal.ctypes.data_as(vp), ctypes.c_double(topo.max_dist), ctypes.c_double(topo.tol),
        ctypes.c_int(1 if rebuild else 0), ctypes.c_int(cap), ctypes.c_int(cap), ctypes.byref(n_mol),
        ctypes.byref(status), off.ctypes.data_as(vp), src.ctypes.data_as(vp), img.ctypes.data_as(vp),
        oxyz.ctypes.data_as(vp), ctypes.c_int(int(with_bits)))
    assert rc == 0
    return RB.molecules_from_output(system, n_mol.value, off, src, oxyz), status.value


@pytest.mark.parametrize("name", SMALL + ["MIBQAR"])
def test_host_team_matches_reference(hostsim, name):
    system, expect = CASES[name]
    mols, status = run_hostsim(hostsim, system, False)
    assert status == 0
    check_molecules(mols, expect, "plain", f"hostsim/{name}")
    if "rebuild_offset" in expect:
        mols, status = run_hostsim(hostsim, system, True)
        assert status == 0
        check_molecules(mols, expect, "rebuild", f"hostsim/{name}/rebuild")
        # cells too large for the visit bit sets in team-shared memory use the stamp arrays
        mols, status = run_hostsim(hostsim, system, True, with_bits=3)      # + scan coordinates in fast memory
        assert status == 0
        check_molecules(mols, expect, "rebuild", f"hostsim/{name}/rebuild/fast-scan")
        mols, status = run_hostsim(hostsim, system, True, with_bits=False)
        assert status == 0
        check_molecules(mols, expect, "rebuild", f"hostsim/{name}/rebuild/stamps")


def load_limits():
    """tests/golden/rebuild_limits.npz: the reference's own output on synth.threshold_cell() - carbon pairs 3e-7 and
    1e-8 either side of the limits of its bond test (Rcov sum -+ tol, max_dist), along an axis, along diagonals and
    through a cell face (make_golden.py rebuild_limits)."""
    from pywindow_amd import synth

    g = np.load(GOLDEN / "rebuild_limits.npz")
    system = {k.split("__in_")[1]: g[k] for k in g.files if k.startswith("limits__in_")}
    expect = {k.split("__")[1]: g[k] for k in g.files if k.startswith("limits__") and "__in_" not in k}
    fresh = synth.threshold_cell()                      # the fixture is what the generator makes today
    assert np.array_equal(fresh["coordinates"], system["coordinates"]) and list(fresh["elements"]) == list(system["elements"])
    return system, expect


def test_pairs_at_the_limits_of_the_bond_test_host(hostsim):
    """The pairs whose bond test cannot be made once per frame are tested where the walk meets them:
    oracle and host-compiled kernel source (every memory layout) against the reference's output."""
    from oracle import pw_rebuild as R

    system, expect = load_limits()
    n_plain, n_reb = len(expect["plain_offset"]) - 1, len(expect["rebuild_offset"]) - 1
    assert 2 < n_reb < n_plain < len(system["elements"])           # some pairs bond, some do not, some only by image
    check_molecules(R.discrete_molecules(system), expect, "plain", "oracle/limits")
    check_molecules(R.discrete_molecules(system, rebuild=R.create_supercell(system)), expect, "rebuild", "oracle/limits/rebuild")
    for bits in (1, 3, 0):
        mols, status = run_hostsim(hostsim, system, False, with_bits=bits)
        assert status == 0
        check_molecules(mols, expect, "plain", f"hostsim/{bits}/limits")
        mols, status = run_hostsim(hostsim, system, True, with_bits=bits)
        assert status == 0
        check_molecules(mols, expect, "rebuild", f"hostsim/{bits}/limits/rebuild")


@pytest.mark.gpu
def test_pairs_at_the_limits_of_the_bond_test_hip(hip_ctx):
    system, expect = load_limits()
    check_molecules(RB.discrete_molecules(dict(system)), expect, "plain", "hip/limits")
    check_molecules(RB.discrete_molecules(dict(system), rebuild=True), expect, "rebuild", "hip/limits/rebuild")


@pytest.mark.gpu
@pytest.mark.parametrize("name", SMALL + ["cc3_cell_md1", "MIBQAR"])
def test_hip_matches_reference(hip_ctx, name):
    system, expect = CASES[name]
    check_molecules(RB.discrete_molecules(dict(system)), expect, "plain", f"hip/{name}")
    if "rebuild_offset" in expect:
        check_molecules(RB.discrete_molecules(dict(system), rebuild=True), expect, "rebuild", f"hip/{name}/rebuild")


@pytest.mark.gpu
def test_hip_many_frames_one_launch(hip_ctx):
    """Frames of one topology in one launch (BASELINE config 4 shape): each frame's result equals
    the single-frame result; frame order does not matter."""
    for names in (["cc3_cell", "cc3_cell_centred"], ["cc3_cell_md0", "cc3_cell_md1"]):
        systems = [CASES[k][0] for k in names]
        assert np.array_equal(systems[0]["elements"], systems[1]["elements"])
        topo = RB.CellTopology(systems[0]["elements"])
        reps = 300
        coords = np.array([s["coordinates"] for s in systems] * reps)
        lat = np.array([s["lattice"] for s in systems] * reps)
        n_mol, off, src, img, xyz = RB.discrete_molecules_frames(topo, coords, lat, True)
        for f in range(len(coords)):
            system, expect = CASES[names[f % len(names)]]
            mols = RB.molecules_from_output(system, n_mol[f], off[f], src[f], xyz[f])
            check_molecules(mols, expect, "rebuild", f"hip/frame{f}")


@pytest.mark.gpu
def test_molecular_system_api(hip_ctx):
    """MolecularSystem.rebuild_system / make_modular as the reference's tests use them
    (tests/test_molecular.py:4467-4553)."""
    import pywindow_amd as pw

    system, expect = CASES["cc3_cell"]
    ms = pw.MolecularSystem.load_system(dict(system), "periodic")
    ms.make_modular()
    assert len(ms.molecules) == 33
    ms.make_modular(rebuild=True)
    assert len(ms.molecules) == 8 and all(m.no_of_atoms == 168 for m in ms.molecules.values())
    rebuilt = ms.rebuild_system()
    assert np.array_equal(rebuilt.system["coordinates"], expect["rebuild_xyz"])
    assert np.array_equal(rebuilt.system["elements"], expect["rebuild_elements"])
    assert np.array_equal(rebuilt.system["atom_ids"], expect["rebuild_ids"])
    rebuilt.make_modular()
    assert len(rebuilt.molecules) == 8
    # the rebuilt cages go straight into the analysis
    props = ms.molecules[0].full_analysis()
    assert props["windows"]["diameters"] is not None and len(props["windows"]["diameters"]) == 4


# ---- periodic DL_POLY trajectory: frames -> rebuilt cages -> full analysis ----------------------
def write_periodic_history(tmp_path):
    from pywindow_amd import synth

    g = np.load(GOLDEN / "ptraj.npz")
    path = tmp_path / "HISTORY_periodic"
    path.write_text(synth.history_text(g["elements"], list(g["frames"]),
                                       title="periodic CC3 cell (pywindow_amd.synth)", cell=g["cell"]))
    return g, path


def test_periodic_history_is_parsed_like_the_reference(tmp_path):
    """Native parser on an imcon=1 HISTORY: coordinates and lattice (cell vectors as columns,
    reference trajectory.py:724-726) equal what the reference's parser produced."""
    import pywindow_amd as pw

    g, path = write_periodic_history(tmp_path)
    traj = pw.DLPOLY(path)
    assert traj.no_of_frames == 2 and traj.no_of_atoms == 1344 and traj.periodic_boundary == "cubic"
    lat = np.zeros((2, 3, 3))
    xyz = traj.read_coordinates(0, 2, lat)
    assert np.array_equal(xyz, g["parsed_coordinates"])
    assert np.array_equal(lat, g["parsed_lattice"])


@pytest.mark.gpu
def test_periodic_trajectory_modular_rebuild(hip_ctx, tmp_path):
    """DLPOLY.analysis(modular=True, rebuild=True) (reference trajectory.py:496-522, the
    Example-8 flow): every frame is rebuilt on the GPU and every cage analysed, all in two
    launches; compared with the reference's output for the same file."""
    import pywindow_amd as pw
    from _util import rel

    g, path = write_periodic_history(tmp_path)
    traj = pw.DLPOLY(path)
    traj.analysis(modular=True, rebuild=True, forcefield="opls")
    cols = list(g["columns"])
    worst = 0.0
    for f in range(2):
        ref = g[f"frame{f}"]
        out = traj.analysis_output[f]
        assert sorted(out) == list(range(len(ref)))
        for row in ref:
            p = out[int(row[0])]
            assert p["no_of_atoms"] == int(row[cols.index("n_atoms")])
            assert np.array_equal(p["centre_of_mass"], row[2:5])
            assert p["maximum_diameter"]["diameter"] == row[cols.index("maxd")]
            assert p["average_diameter"] == row[cols.index("avg_d")]
            assert p["pore_diameter"]["diameter"] == row[cols.index("pore_d")]
            assert p["pore_diameter_opt"]["diameter"] == row[cols.index("pore_opt_d")]
            nw = int(row[cols.index("n_windows")])
            assert len(p["windows"]["diameters"]) == nw
            e = rel(np.sort(p["windows"]["diameters"]), row[10:10 + nw])
            assert e == 0.0
            worst = max(worst, e)
    print("periodic trajectory: worst window rel err", worst)
    # columnar form, and a frame the reference cannot finish is still analysed here
    recs, uframe, umol = traj.modular_records(frames=[1], rebuild=True, forcefield="opls")
    assert len(recs) == 8 and (uframe == 1).all() and list(umol) == list(range(8))


@pytest.mark.gpu
def test_long_modular_analysis_in_pieces(hip_ctx, tmp_path, monkeypatch):
    """A very long periodic trajectory goes through the device in pieces: same records, same
    order as the one-piece analysis."""
    import pywindow_amd as pw
    from pywindow_amd import synth, trajectory

    g = np.load(GOLDEN / "ptraj.npz")
    rng = np.random.default_rng(11)
    base = np.asarray(g["frames"][0], float)
    frames = [base + rng.normal(0.0, 0.01, size=base.shape) for _ in range(11)]
    path = tmp_path / "HISTORY_periodic_long"
    path.write_text(synth.history_text(g["elements"], frames, cell=g["cell"]))
    traj = pw.DLPOLY(path)
    whole = traj.modular_records(rebuild=True, forcefield="opls")
    monkeypatch.setattr(trajectory, "MODULAR_CHUNK", 2)
    monkeypatch.setattr(trajectory, "MODULAR_PIECE", 2)      # 11 frames -> six pieces, three of them in flight at a time
    parts = traj.modular_records(rebuild=True, forcefield="opls")
    assert len(whole[0]) >= 11 * 8
    assert whole[0].tobytes() == parts[0].tobytes()
    assert np.array_equal(whole[1], parts[1]) and np.array_equal(whole[2], parts[2])


@pytest.mark.gpu
@pytest.mark.parametrize("copies", [(2, 1, 1), (2, 2, 1), (2, 2, 2)], ids=["2688 atoms", "5376 atoms", "10752 atoms"])
def test_hip_larger_cells_take_the_other_memory_layouts(hip_ctx, hostsim, copies):
    """Copies of the CC3 test cell side by side: 2 688 atoms (visit bit sets in team-shared memory, candidate grid and
    lists in the team's slab, one-wave walk over global lists), 5 376 (the same, near the limit of the bit sets)
    and 10 752 (no bit sets: stamp arrays and the four-wave layer loop) - each against the host-compiled kernel
    source, whose agreement with the reference the smaller cases pin."""
    from pywindow_amd import rebuild as RB

    _larger_cell_check(hostsim, "cc3_cell", copies, 8)


@pytest.mark.gpu
def test_hip_framework_in_a_larger_cell(hip_ctx, hostsim):
    """2 x 2 x 1 copies of the framework that fills its supercell (1 696 atoms, one molecule of 45 792): breadth-first
    layers far wider than the lists kept in team-shared memory, walked by one wave with the list tails in the
    team's slab."""
    _larger_cell_check(hostsim, "MIBQAR", (2, 2, 1), None)


def _larger_cell_check(hostsim, case, copies, mols_per_cell):
    from pywindow_amd import rebuild as RB

    base = CASES[case][0]
    lat = np.asarray(base["lattice"], float)
    xyz0 = np.asarray(base["coordinates"], float)
    shifts = [(a, b, c) for a in range(copies[0]) for b in range(copies[1]) for c in range(copies[2])]
    xyz = np.concatenate([xyz0 + lat @ np.array(s, float) for s in shifts])
    big = lat * np.array(copies, float)[None, :]            # (columns are the cell vectors)
    system = {"elements": np.concatenate([np.asarray(base["elements"])] * len(shifts)), "coordinates": xyz, "lattice": big}
    topo = RB.CellTopology(system["elements"])
    n_mol, off, src, img, out = RB.discrete_molecules_frames(topo, xyz[None], big[None], True)
    got = RB.molecules_from_output(system, int(n_mol[0]), off[0], src[0], out[0])
    want, status = run_hostsim(hostsim, system, True, with_bits=False)
    assert status == 0 and len(got) == len(want)
    assert mols_per_cell is None or len(got) == mols_per_cell * len(shifts)
    for g, w in zip(got, want):
        assert list(g["elements"]) == list(w["elements"])
        assert np.array_equal(g["coordinates"], w["coordinates"])


@pytest.mark.gpu
def test_resident_hand_over_matches_host_path(hip_ctx):
    """pw_resident_from_cells: the ragged unit batch built on the device gives the same analysis
    records as the host-marshalled path; capacity retries (a framework 27x the cell) work."""
    from pywindow_amd import _lib
    from pywindow_amd import element_data as E

    for name, rebuild in (("cc3_cell_md1", True), ("cc3_cell", False), ("EPIRUR", True)):
        system = CASES[name][0]
        topo = RB.CellTopology(system["elements"])
        ids = E.element_ids(system["elements"])
        coords, lat, inv = RB.pack_frames(np.array([system["coordinates"]] * 3), np.array([system["lattice"]] * 3))
        res, n_mol = hip_ctx.resident_from_cells(topo, E.VDW[ids], coords, lat, inv, rebuild)
        mols = RB.discrete_molecules(dict(system), rebuild=True if rebuild else None)
        assert list(n_mol) == [len(mols)] * 3 and res.n_units == 3 * len(mols)
        if max(len(m["elements"]) for m in mols) >= 20:
            res.launch(_lib.STAGE_BASIC | _lib.STAGE_AVG)
            got = res.download()
            from pywindow_amd import engine

            want = engine.analyse([(m["elements"], m["coordinates"]) for m in mols], _lib.STAGE_BASIC | _lib.STAGE_AVG)
            for k in range(3):
                part = got[k * len(mols):(k + 1) * len(mols)]
                for key in ("n_atoms", "mw", "com", "maxd", "pore_d", "avg_d"):
                    assert np.array_equal(part[key], want[key]), (name, key)
        res.free()
    system = CASES["MIBQAR"][0]
    topo = RB.CellTopology(system["elements"])
    ids = E.element_ids(system["elements"])
    coords, lat, inv = RB.pack_frames(system["coordinates"][None], system["lattice"][None])
    res, n_mol = hip_ctx.resident_from_cells(topo, E.VDW[ids], coords, lat, inv, True)
    assert list(n_mol) == [1] and res.n_units == 1
    res.free()
