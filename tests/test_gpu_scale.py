"""BASELINE.json's large shapes at single-GPU full size, through the product API and the C ABI:

* config 4 -- a >= 1000-frame periodic DL_POLY trajectory (8 CC3 cages per cell) from the HISTORY file to
  the records: native parse, periodic re-assembly of every frame and analysis of every cage
  (``DLPOLY.modular_records(rebuild=True)``, the Example-8 flow, reference trajectory.py:496-522),
  a sample of cages against the oracle run live, run-to-run and order independence;
* config 4 / 5 unit counts (80 000 and 500 000 units in ONE launch): size-independent properties.
"""
import numpy as np
import pytest

from _util import GOLDEN, LIVE_TOL_WINDOW, ROOT, rel

pytestmark = pytest.mark.gpu

N_FRAMES = 1024


def _cell():
    g = np.load(GOLDEN / "rebuild.npz")
    return g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]


@pytest.fixture(scope="module")
def periodic_history(tmp_path_factory):
    """1024 frames of the 1344-atom cubic CC3 cell (tests/data/system_periodic.pdb of the reference, the
    input of its rebuild test) + N(0, 0.02 A) noise, frame k from default_rng(4 + k), as a DL_POLY
    HISTORY file (imcon = 1)."""
    from pywindow_amd import synth

    el, xyz, lat = _cell()
    cell = np.asarray(lat, float).T            # rows = cell vectors in the file
    frames = (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(N_FRAMES))
    path = tmp_path_factory.mktemp("config4") / "HISTORY_periodic_1024"
    synth.write_history(path, el, frames, title="periodic CC3 cell x 1024 frames (tests/test_gpu_scale.py)", cell=cell)
    return path


def test_config4_periodic_trajectory_end_to_end(hip_ctx, periodic_history, monkeypatch):
    import pywindow_amd as pw
    from oracle import pw_oracle as O
    from oracle import pw_rebuild as R
    from pywindow_amd import element_data as E
    from pywindow_amd import rebuild as rb

    traj = pw.DLPOLY(periodic_history)
    assert (traj.no_of_frames, traj.no_of_atoms, traj.periodic_boundary) == (N_FRAMES, 1344, "cubic")
    recs, uframe, umol = traj.modular_records("all", rebuild=True)
    # 8 cages per frame, in frame order then molecule order; every cage whole and analysed
    assert len(recs) == 8 * N_FRAMES
    assert np.array_equal(uframe, np.repeat(np.arange(N_FRAMES), 8))
    assert np.array_equal(umol, np.tile(np.arange(8), N_FRAMES))
    assert (recs["status"] == 0).all()
    assert (recs["n_atoms"] == 168).all()
    assert (recs["n_windows"] > 0).all() and (recs["n_windows"] == 4).sum() > 0.99 * len(recs)
    # run to run: byte-identical -- the second time in four pieces of 256 frames (the pipelined form long
    # trajectories take: the next piece is read and re-assembled while the previous ones are analysed)
    from pywindow_amd import trajectory

    monkeypatch.setattr(trajectory, "MODULAR_PIECE", 256)
    again, uf2, um2 = traj.modular_records("all", rebuild=True)
    monkeypatch.undo()
    assert again.tobytes() == recs.tobytes()
    assert np.array_equal(uf2, uframe) and np.array_equal(um2, umol)
    # frame order does not matter (a permuted selection gives the permuted records)
    perm = np.random.default_rng(1).permutation(N_FRAMES)[:96].tolist()
    sub, sf, sm = traj.modular_records(perm, rebuild=True)
    assert np.array_equal(sf, np.repeat(np.array(perm), 8))
    want = np.concatenate([recs[8 * f: 8 * f + 8] for f in perm])
    assert sub.tobytes() == want.tobytes()
    # a sample of 32 cages against the oracle run live: the re-assembled molecule (atom order and
    # coordinates) from the product API, then every scalar of its analysis
    rng = np.random.default_rng(2)
    frames = rng.choice(N_FRAMES, 4, replace=False)
    el = traj.elements()
    checked = 0
    for f in frames:
        lat = np.zeros((1, 3, 3))
        xyz = traj.read_coordinates(int(f), 1, lat)[0]
        system = {"elements": el, "atom_ids": traj.atom_ids, "coordinates": xyz, "lattice": lat[0],
                  "unit_cell": rb.lattice_array_to_unit_cell(lat[0])}
        mine = rb.discrete_molecules(dict(system), rebuild=True)
        if checked == 0:       # the periodic re-assembly itself, one frame against the oracle's restatement
            theirs = R.discrete_molecules(system, rebuild=R.create_supercell(system))
            assert len(mine) == len(theirs) == 8
            for a, b in zip(mine, theirs):
                assert np.array_equal(a["coordinates"], b["coordinates"]) and np.array_equal(a["elements"], b["elements"])
        for m, mol in enumerate(mine):
            ids = E.element_ids(mol["elements"])
            ref = O.full_analysis(mol["coordinates"], E.VDW[ids], E.MASS[ids])
            r = recs[8 * int(f) + m]
            for key in ("mw", "maxd", "avg_d", "pore_d", "pore_opt_d"):
                assert float(r[key]) == ref[key], (f, m, key)
            assert (int(r["maxd_i"]), int(r["maxd_j"])) == (ref["maxd_i"], ref["maxd_j"])
            assert np.array_equal(r["pore_opt_c"], ref["pore_opt_c"])
            nw = ref["n_windows"]
            assert int(r["n_windows"]) == nw > 0
            assert rel(r["win_d"][:nw], ref["win_d"][:nw]) <= LIVE_TOL_WINDOW
            checked += 1
    assert checked == 32
    # the nested-dict API on the same file: keys as the reference's modular branch builds them
    traj.analysis(frames=[0, N_FRAMES - 1], modular=True, rebuild=True)
    assert sorted(traj.analysis_output) == [0, N_FRAMES - 1]
    assert sorted(traj.analysis_output[0]) == list(range(8))
    p = traj.analysis_output[N_FRAMES - 1][7]
    assert p["pore_diameter_opt"]["diameter"] == recs[-1]["pore_opt_d"]
    assert np.array_equal(p["windows"]["diameters"], recs[-1]["win_d"][: int(recs[-1]["n_windows"])])


@pytest.mark.parametrize("units,label", [(80_000, "config 4: 10 000 frames x 8 cages"),
                                         (500_000, "config 5: 5000 cages x 100 frames")])
def test_full_unit_counts_in_one_launch(hip_ctx, units, label):
    """Every unit of the configuration resident at once, one launch: all finish with status 0, and a
    random sample of 256 re-analysed on its own (another batch size, another launch plan, other
    neighbours) is byte-identical -- results do not depend on what else is in the batch."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, base = synth.load_cc3_base()
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    rng = np.random.default_rng(units)
    coords = np.empty((units,) + base.shape)
    for s in range(0, units, 20000):
        e = min(units, s + 20000)
        coords[s:e] = base[None] + rng.normal(0.0, 0.10, size=(e - s,) + base.shape)
    res = hip_ctx.upload(_lib.Batch.uniform(coords, vdw, mass))
    res.launch()
    recs = res.download()
    res.free()
    assert len(recs) == units, label
    assert (recs["status"] == 0).all(), label
    assert (recs["n_atoms"] == 168).all()
    assert (recs["n_windows"] == 4).sum() > 0.95 * units
    assert (recs["pore_opt_d"] >= recs["pore_d"]).all()          # the optimiser never makes the pore smaller
    pick = np.sort(rng.choice(units, 256, replace=False))
    small = hip_ctx.analyse(_lib.Batch.uniform(coords[pick], vdw, mass))
    assert small.tobytes() == recs[pick].tobytes(), label


def test_config4_ten_thousand_frame_file_on_one_gpu(hip_ctx, monkeypatch):
    """BASELINE configs[3] at its full length through the FILE path on one GPU (round-5 review: the file path had only
    run at 1024 frames): a 10 000-frame periodic HISTORY (8 CC3 cages / 1344 atoms per cell; 64 distinct noisy frames
    cycled -- 1.08 GB of text in shared memory, written in a quarter of a second) tokenised, re-assembled and analysed
    by DLPOLY.modular_records(rebuild=True) in twenty pieces.  Every cage status 0 with its four windows; the first
    1024 frames byte-identical to the same frames analysed on their own (two pieces, the interleaved schedule); and,
    the frames being a cycle of 64, every block of 64 frames byte-identical to the first -- 156 independent repetitions
    of the same 512 analyses in other pieces, beside other neighbours."""
    import os
    import tempfile

    import pywindow_amd as pw
    from pywindow_amd import synth

    g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
    el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    frames = 10_000
    with tempfile.TemporaryDirectory(dir=base) as tmp:
        path = os.path.join(tmp, "HISTORY_periodic")
        distinct = [xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(64)]
        synth.write_history_cycled(path, el, distinct, frames, cell=np.asarray(lat, float).T)
        traj = pw.DLPOLY(path)
        assert (traj.no_of_frames, traj.no_of_atoms) == (frames, 1344)
        recs, uframe, umol = traj.modular_records("all", rebuild=True)
        legs = dict(traj.last_timings)
        assert legs["pieces"] == 20 and legs.get("groups", 0) >= 1, legs
        assert len(recs) == 8 * frames
        assert np.array_equal(uframe, np.repeat(np.arange(frames), 8)) and np.array_equal(umol, np.tile(np.arange(8), frames))
        assert (recs["status"] == 0).all() and (recs["n_atoms"] == 168).all()
        assert (recs["n_windows"] == 4).sum() > 0.99 * len(recs)
        head, _, _ = traj.modular_records(list(range(1024)), rebuild=True)
        assert traj.last_timings["pieces"] == 2
        assert head.tobytes() == recs[: 8 * 1024].tobytes()
        block = recs[: 8 * 64].tobytes()
        for k in range(1, frames // 64):
            assert recs[8 * 64 * k: 8 * 64 * (k + 1)].tobytes() == block, k
        # the schedule of rounds 1-5 (every piece analysed at once, the next re-assembled beside it): the same bytes
        monkeypatch.setenv("PW_MODULAR_GROUP", "0")
        old, _, _ = traj.modular_records(list(range(3072)), rebuild=True)
        assert old.tobytes() == recs[: 8 * 3072].tobytes()
