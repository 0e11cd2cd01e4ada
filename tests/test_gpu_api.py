"""The reference-shaped API (Molecule / functions / DLPOLY.analysis) on the GPU."""
import numpy as np
import pytest

from _util import ROOT, load_group, molecules, rel

pytestmark = pytest.mark.gpu


def test_molecule_full_analysis_schema_and_values():
    import pywindow_amd as pw

    g = load_group("static")
    el, xyz = molecules(g)[0]
    ms = pw.MolecularSystem.load_system({"elements": el, "coordinates": xyz}, "test")
    mol = ms.system_to_molecule()
    p = mol.full_analysis()
    assert list(p) == ["no_of_atoms", "centre_of_mass", "maximum_diameter", "average_diameter",
                       "pore_diameter", "pore_volume", "pore_diameter_opt", "pore_volume_opt", "windows"]
    assert p["no_of_atoms"] == 168 and isinstance(p["no_of_atoms"], int)
    assert p["maximum_diameter"] == {"diameter": 22.179369990077188, "atom_1": 12, "atom_2": 54}
    assert p["pore_diameter"] == {"diameter": 5.397020177310022, "atom": 29}
    assert isinstance(p["centre_of_mass"], np.ndarray) and p["windows"]["centre_of_mass"].shape == (4, 3)
    assert mol.MW == 1117.5479999999998
    # method-by-method calls agree with the one-launch path
    mol2 = ms.system_to_molecule()
    assert mol2.calculate_maximum_diameter() == p["maximum_diameter"]["diameter"]
    assert mol2.calculate_pore_volume() == p["pore_volume"]
    assert mol2.calculate_average_diameter() == p["average_diameter"]
    assert mol2.calculate_pore_diameter_opt() == p["pore_diameter_opt"]["diameter"]
    assert np.array_equal(mol2.calculate_windows(), p["windows"]["diameters"])
    mol2.shift_to_origin()
    assert np.max(np.abs(mol2.calculate_centre_of_mass())) < 1e-12


def test_no_window_molecule_gives_none():
    import pywindow_amd as pw

    g = load_group("static")
    el, xyz = molecules(g)[1]   # C60, reference tests/test_validate_windows.py:2001-2007
    mol = pw.Molecule({"elements": el, "coordinates": xyz}, "c60", 0)
    assert mol.calculate_windows() is None
    assert mol.properties["windows"] == {"diameters": None, "centre_of_mass": None}
    assert pw.find_windows(el, xyz) is None


def test_function_level_mirror():
    import pywindow_amd as pw

    g = load_group("md20")
    el, xyz = molecules(g)[3]
    assert pw.molecular_weight(el) == g["mw"][3]
    assert np.array_equal(pw.center_of_mass(el, xyz), g["com"][3])
    assert pw.max_dim(el, xyz) == (int(g["maxd_i"][3]), int(g["maxd_j"][3]), float(g["maxd"][3]))
    assert pw.pore_diameter(el, xyz) == (float(g["pore_d"][3]), int(g["pore_atom"][3]))
    d, a, c = pw.opt_pore_diameter(el, xyz)
    assert d == g["pore_opt_d"][3] and a == g["pore_opt_atom"][3] and np.array_equal(c, g["pore_opt_c"][3])
    assert pw.pore_diameter(el, xyz, com=c) == (d, a)
    assert pw.find_average_diameter(el, xyz) == g["avg_d"][3]
    wd, wc = pw.find_windows(el, xyz)
    assert rel(np.sort(wd), np.sort(g["win_d"][3][:4])) == 0.0
    with pytest.raises(KeyError):
        pw.molecular_weight(np.array(["Qq"]))


def test_dlpoly_batched_analysis(tmp_path):
    import pywindow_amd as pw
    from pywindow_amd import synth

    path = synth.write_synthetic_history(tmp_path / "HISTORY_synth", 12)
    traj = pw.DLPOLY(path)
    assert traj.no_of_frames == 12 and traj.no_of_atoms == 168
    traj.analysis(frames=[0, 1, 2, 5])
    assert sorted(traj.analysis_output) == [0, 1, 2, 5]
    traj.analysis()      # the rest; already analysed frames are skipped (override=False)
    assert sorted(traj.analysis_output) == list(range(12))
    g = load_group("synth64")
    for f in range(12):
        p = traj.analysis_output[f]["0"]
        assert p["pore_diameter_opt"]["diameter"] == g["pore_opt_d"][f]
        assert p["average_diameter"] == g["avg_d"][f]
        assert len(p["windows"]["diameters"]) == g["n_windows"][f]
    with pytest.raises(Exception):
        traj.analysis(frames="bogus")
    # a long selection is analysed in pieces (parsing overlaps the kernels): same records, same order
    from pywindow_amd import trajectory

    whole = traj.analysis_records()
    old = trajectory.RUN_PIECE
    trajectory.RUN_PIECE = 4
    try:
        assert traj.analysis_records().tobytes() == whole.tobytes()       # 12 frames -> 3 pieces of 4
        assert traj.analysis_records(frames=list(range(11))).tobytes() == whole[:11].tobytes()
    finally:
        trajectory.RUN_PIECE = old


def test_record_gather_over_rccl_single_rank():
    """The RCCL (nccl backend) path of the only collective, with one rank on the GPU."""
    import os

    import torch
    import torch.distributed as dist

    from pywindow_amd import _lib
    from pywindow_amd.trajectory import gather_records

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        local = np.zeros(5, dtype=_lib.UNIT_OUT_DTYPE)
        local["pore_d"] = np.arange(5) * 1.5
        local["n_windows"] = 4
        out = gather_records(local, 5, 0, 1, dist)
        assert out.tobytes() == local.tobytes()
    finally:
        dist.destroy_process_group()


def test_contexts_batches_and_overlapped_launches(hip_ctx):
    """Odd batch sizes on fresh contexts, two resident batches alternating on one context
    (overlapped pipeline launches), single-stage launches in between."""
    import runpy
    import pathlib

    runpy.run_path(str(pathlib.Path(__file__).resolve().parent / "tools" / "stress.py"), run_name="__main__")


def test_torch_can_start_after_the_library():
    """A process that analyses first and touches torch.cuda afterwards (the library loads before
    torch): both must end up on one HIP runtime, or torch finds no GPU.  Starts a second Python
    process that pages torch in again (minutes on a cold box), so it only runs on request:
    PW_TEST_TORCH_ORDER=1 (or run tests/tools/torch_after.py by hand)."""
    import os
    import pathlib
    import subprocess
    import sys

    if os.environ.get("PW_TEST_TORCH_ORDER") != "1":
        pytest.skip("set PW_TEST_TORCH_ORDER=1 to run")

    tool = pathlib.Path(__file__).resolve().parent / "tools" / "torch_after.py"
    out = subprocess.run([sys.executable, str(tool), "both"], capture_output=True, text=True, timeout=1200,   # (a fresh box pages torch in for minutes)
                         cwd=str(tool.parents[2]))
    assert out.returncode == 0 and "torch ok" in out.stdout, out.stdout[-500:] + out.stderr[-1500:]


def test_stream_probe_is_stable():
    """pw_context_create measures whether the pipeline's ten streams run concurrently and falls back to single
    launches when they do not.  In a process of its own, with GPU_MAX_HW_QUEUES exported before HIP starts
    (importing the package first does that) and one context alive at a time, the answer must be "they do",
    every time: a probe that misfires would silently cost 2.6x of the throughput.  (Several contexts alive on
    one device share the hardware queues; the later ones then run single launches -- this process has a few.)"""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from pywindow_amd import _lib\n"
        "bad = []\n"
        "for k in range(25):\n"
        "    ctx = _lib.Context(0)\n"
        "    if not ctx.pipelined: bad.append(k)\n"
        "    ctx.close()\n"
        "print('BAD', bad)\n"
    ) % str(ROOT)
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-1500:]
    assert "BAD []" in proc.stdout, (proc.stdout[-500:], proc.stderr[-1500:])


def test_pacing_gates_do_not_time_out(hip_ctx):
    """The pipeline's gates only pace launches; one that gives up waiting costs 20 ms and is counted."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(96)
    ids = E.element_ids(elements)
    res = hip_ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
    for _ in range(6):
        res.launch()
    res.download()
    res.free()
    assert hip_ctx.gate_timeouts == {"tail": 0, "head": 0, "residency": 0}


@pytest.mark.gpu
def test_threads_on_one_context_and_on_two_contexts():
    """The ABI's threading contract (include/pywindow_amd.h, "Threads"; the reference's workers are
    stateless processes, trajectory.py:564-582): calls from several threads on ONE context serialise inside the
    library, two contexts share no state -- the second live context of a device runs its analyses as single
    launches -- and every thread gets the records a lone thread gets, byte for byte.  Trajectories analysed from
    two threads through the shared per-device context keep their own frames (the staging buffer and the "records
    fetched last" list are held under Context.lock)."""
    import threading

    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(96)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    parts = [frames[:48], frames[48:]]
    first = _lib.Context(0)
    expect = [first.analyse(_lib.Batch.uniform(p, vdw, mass)) for p in parts]
    assert first.pipelined

    def run_threads(ctxs, rounds=3):
        got, errors = [None, None], []

        def work(k):
            try:
                ctx = ctxs[k]
                for _ in range(rounds):
                    # the one-call path ...
                    a = ctx.analyse(_lib.Batch.uniform(parts[k], vdw, mass))
                    # ... and the resident path, whose calls interleave with the other thread's
                    res = ctx.upload(_lib.Batch.uniform(parts[k], vdw, mass))
                    res.launch()
                    b = res.download_settled()
                    res.free()
                    assert a.tobytes() == b.tobytes()
                    got[k] = b
            except Exception as exc:  # noqa: BLE001
                errors.append(repr(exc))

        ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=600)
        assert not errors, errors
        return got

    # two threads, ONE context
    got = run_threads([first, first])
    assert got[0].tobytes() == expect[0].tobytes() and got[1].tobytes() == expect[1].tobytes()
    # two threads, a context each on the same device (the second one finds the hardware queues taken and runs
    # every analysis as a single launch: correct, slower)
    second = _lib.Context(0)
    got = run_threads([first, second])
    assert got[0].tobytes() == expect[0].tobytes() and got[1].tobytes() == expect[1].tobytes()
    second.close()
    first.close()


@pytest.mark.gpu
def test_two_threads_analyse_trajectories_on_the_shared_context(tmp_path):
    """DLPOLY._run decodes into the context's ONE page-locked buffer and reads windows beyond
    W_MAX from the context's "fetched last" list; two threads analysing different trajectories on the same
    device must not see each other's frames."""
    import threading

    import pywindow_amd as pw
    from pywindow_amd import synth

    paths = [synth.write_synthetic_history(tmp_path / f"HISTORY_{k}", 200, seed_base=synth.SEED_BASE + 1000 * k) for k in range(2)]
    alone = []
    for p in paths:
        t = pw.DLPOLY(p)
        alone.append(t.analysis_records(forcefield="opls", swap_atoms={"he": "H"}))
    assert alone[0].tobytes() != alone[1].tobytes()
    got, errors = [None, None], []

    def work(k):
        try:
            for _ in range(4):
                got[k] = pw.DLPOLY(paths[k]).analysis_records(forcefield="opls", swap_atoms={"he": "H"})
                assert got[k].tobytes() == alone[k].tobytes()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not errors, errors


@pytest.mark.gpu
def test_streamed_batch_equals_the_uploaded_one(tmp_path):
    """pw_resident_stream_begin / _append: the analysis is launched BEFORE any coordinate is on the device and the
    reader feeds it; records byte-identical to the batch uploaded in one piece, whatever the chunking, also when the
    launch is repeated on the completed batch; an incomplete batch cannot be downloaded; appends must be in order.
    And the trajectory driver built on it: DLPOLY.analysis_records streams a 1000-frame file (last_timings says so)
    and returns what the one-piece path returns."""
    import pywindow_amd as pw
    from pywindow_amd import _lib, engine, synth, trajectory
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(300)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    ctx = engine.context(0)
    whole = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
    whole.launch()
    expect = whole.download()
    whole.free()
    for chunk in (300, 64, 7):
        buf = ctx.pinned_array(frames.shape)
        buf[:] = frames
        res = ctx.stream_begin(len(frames), vdw, mass)
        res.launch()                                   # nothing has arrived yet: the teams wait
        with pytest.raises(_lib.PwHipError):
            res.download()                             # incomplete
        for lo in range(0, len(frames), chunk):
            res.append(buf[lo:lo + chunk])
        got = res.download()
        assert got.tobytes() == expect.tobytes(), chunk
        res.launch()                                   # a second analysis of the complete batch
        assert res.download().tobytes() == expect.tobytes()
        with pytest.raises(_lib.PwHipError):
            res.append(buf[:1])                        # nothing left to append
        res.free()
    path = synth.write_synthetic_history(tmp_path / "HISTORY", 1000)
    traj = pw.DLPOLY(path)
    streamed = traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    assert traj.last_timings["streamed"] is True and traj.last_timings["pieces"] == 1
    old = trajectory.STREAM_MIN
    trajectory.STREAM_MIN = 10 ** 9
    try:
        plain = pw.DLPOLY(path).analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    finally:
        trajectory.STREAM_MIN = old
    assert streamed.tobytes() == plain.tobytes() and (streamed["n_windows"] == 4).all()
    # consecutive frames went through the native streamed read (pw_history_stream_read: decoded and appended side by
    # side); a selection with a gap takes the chunked appends of the binding, a sub-range the native read again --
    # each the records of those frames, and the native read by hand with small appends and an offset
    gap = [f for f in range(1000) if f != 500][:400] + list(range(600, 1000))
    assert traj.analysis_records(frames=gap, forcefield="opls", swap_atoms={"he": "H"}).tobytes() == plain[gap].tobytes()
    assert traj.last_timings["streamed"] is True
    sub = list(range(123, 123 + 321))
    assert traj.analysis_records(frames=sub, forcefield="opls", swap_atoms={"he": "H"}).tobytes() == plain[sub].tobytes()
    el = traj.elements({"he": "H"}, "opls")
    ids = E.element_ids(el)
    with ctx.lock:
        buf = ctx.pinned_array((300, len(ids), 3))
        res = ctx.stream_begin(300, E.VDW[ids], E.MASS[ids])
        res.launch()
        res.append_from_history(traj._h, 40, buf[:100], 8)          # units 0..99 = frames 40..139
        res.append_from_history(traj._h, 140, buf[100:], 1000)      # the rest in one append
        got = res.download()
        res.free()
    assert got.tobytes() == plain[40:340].tobytes()
    with ctx.lock:
        res = ctx.stream_begin(300, E.VDW[ids], E.MASS[ids])
        with pytest.raises(_lib.PwHipError):
            res.append_from_history(traj._h, 900, buf, 64)            # frames 900..1199 of a 1000-frame file
        res.free()


def test_streamed_batch_on_a_context_without_the_pipeline():
    """A context that runs every analysis as ONE launch (PW_FUSED=1; or the hardware queues were taken by whoever
    initialised the GPU first -- torch before this library without GPU_MAX_HW_QUEUES=12 exported) launches on the stream
    the appends copy on: the launch of a streamed batch is then made by the append that completes it -- found as a
    5 s time-out with torch initialised first.  Also stages without the window search on a pipelined context."""
    import os

    from pywindow_amd import _lib, engine, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(300)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    main = engine.context(0)
    whole = main.upload(_lib.Batch.uniform(frames, vdw, mass))
    whole.launch()
    expect = whole.download()
    whole.free()
    old = os.environ.get("PW_FUSED")
    os.environ["PW_FUSED"] = "1"
    try:
        ctx = _lib.Context(0)
    finally:
        if old is None:
            del os.environ["PW_FUSED"]
        else:
            os.environ["PW_FUSED"] = old
    assert _lib.load().pw_context_pipelined(ctx._h) == 0
    buf = ctx.pinned_array(frames.shape)
    buf[:] = frames
    res = ctx.stream_begin(len(frames), vdw, mass)
    res.launch()
    with pytest.raises(_lib.PwHipError):
        res.download()                                 # incomplete
    for lo in range(0, len(frames), 100):
        res.append(buf[lo:lo + 100])
    assert res.download().tobytes() == expect.tobytes()
    res.launch()
    assert res.download().tobytes() == expect.tobytes()
    res.free()
    ctx.close()
    # the pipelined context, stages that are one launch
    with main.lock:
        buf = main.pinned_array(frames.shape)
        buf[:] = frames
        res = main.stream_begin(len(frames), vdw, mass)
        res.launch(_lib.STAGE_BASIC | _lib.STAGE_OPT)
        for lo in range(0, len(frames), 150):
            res.append(buf[lo:lo + 150])
        got = res.download()
        res.free()
    for k in ("pore_opt_d", "pore_opt_c", "opt_nit", "opt_nfev", "maxd", "pore_d"):
        assert np.array_equal(got[k], expect[k]), k
    # a streamed batch given up half way (a reader that failed): freeing it tells the waiting launches to stop -- no
    # 5 s of waiting for units that never come -- and the context analyses the next batch as if nothing had happened
    import time

    with main.lock:
        buf = main.pinned_array(frames.shape)
        buf[:] = frames
        res = main.stream_begin(len(frames), vdw, mass)
        res.launch()
        res.append(buf[:100])
        t0 = time.perf_counter()
        res.free()
        assert time.perf_counter() - t0 < 1.0
        whole = main.upload(_lib.Batch.uniform(frames, vdw, mass))
        whole.launch()
        assert whole.download().tobytes() == expect.tobytes()
        whole.free()


def test_a_broken_frame_in_a_streamed_trajectory_is_an_error_not_a_stall(tmp_path):
    """A HISTORY whose frame 270 of 300 has a coordinate line that is not numbers: the streamed read reports it as the
    plain read does (_TrajectoryError), at once, and the next analysis on the context is unaffected."""
    import time

    import pywindow_amd as pw
    from pywindow_amd import synth
    from pywindow_amd.trajectory import _TrajectoryError

    good = synth.write_synthetic_history(tmp_path / "GOOD", 300)
    text = open(good).read().split("\n")
    starts = [i for i, line in enumerate(text) if line.startswith("timestep")]
    text[starts[270] + 2] = "   not-a-number   1.0   2.0"          # the first atom's coordinate line of frame 270
    bad = tmp_path / "BAD"
    open(bad, "w").write("\n".join(text))
    expect = pw.DLPOLY(good).analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    t0 = time.perf_counter()
    with pytest.raises(_TrajectoryError):
        pw.DLPOLY(bad).analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    assert time.perf_counter() - t0 < 2.0
    again = pw.DLPOLY(good).analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    assert again.tobytes() == expect.tobytes()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_division_by_a_tabulated_reciprocal_has_the_bits_of_the_division(mode):
    """pw_div_r(a, b, pw_recip_hw(b)) (pw_common.hpp: the solves of the optimisers divide by the same diagonal at every
    step) against a / b on the device, 2^26 operand pairs per class: ordinary magnitudes, arbitrary bit patterns
    (denormals, infinities, NaNs, signed zeros), divisors with an all-ones significand, exact quotients."""
    import ctypes

    from pywindow_amd import _lib

    L = _lib.load()
    L.pw_internal_div_check.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p]
    L.pw_internal_div_check.restype = ctypes.c_int
    out = (ctypes.c_ulonglong * 4)()
    assert L.pw_internal_div_check(0, 1 << 26, mode, 20260000 + mode, out) == 0
    bad = list(out)
    a, b, got = (np.array(bad[1:], dtype=np.uint64).view(np.float64))
    assert bad[0] == 0, f"{bad[0]} quotients differ; first: {a!r} / {b!r} gave {got!r}, want {a / b!r}"


def test_a_path_that_would_never_end_is_flagged_on_the_device(hip_ctx):
    """PW_ST_PATH_TOO_LONG (tests/test_host_context.py has the story): on the device the alternative to the status is a
    launch that never ends.  The unit is flagged within a normal launch and the others are analysed as ever."""
    import time

    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(40)
    ids = E.element_ids(elements)
    batch = _lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids])
    ok = hip_ctx.analyse(batch, _lib.STAGE_ALL)
    t0 = time.perf_counter()
    rec = hip_ctx.analyse(batch, _lib.STAGE_ALL, _lib.Params(increment=1.0e-6))
    assert time.perf_counter() - t0 < 5.0
    assert (rec["status"] & _lib.ST_PATH_TOO_LONG).all() and (rec["n_windows"] == -1).all()
    for k in ("maxd", "avg_d", "pore_d", "pore_opt_d", "pore_opt_c"):
        assert np.array_equal(rec[k], ok[k]), k
    again = hip_ctx.analyse(batch, _lib.STAGE_ALL)
    assert again.tobytes() == ok.tobytes()
