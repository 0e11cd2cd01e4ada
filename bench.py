#!/usr/bin/env python3
"""bench.py -- trajectory frames/s for full_analysis (pore + windows) on the
1000-frame synthetic CC3 trajectory (BASELINE.json config 2), N GPUs of one node.

One *step* = one pass of the whole hot path (all stages) over the rank's batch
of frames, inputs already resident in HBM.  ``--gpus N``:

* under ``torchrun`` (RANK / WORLD_SIZE set) this process is one rank, bound to GPU LOCAL_RANK;
* started plainly with ``--gpus N > 1`` the process starts ``torch.distributed.run`` with N ranks as a
  CHILD (before anything here touches a GPU) and passes its output through.

Weak scaling is the headline (every rank analyses its own 1000 frames per step); with more than one
rank every step ends with the path's only collective, the RCCL all-gather of the result records,
read straight from the engine's device buffer, inside the timed region.  The same run then times
strong scaling (ONE 1000-frame trajectory split over the ranks) and reports it beside the headline.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pipeline keeps several kernels in flight on separate streams: ask the HIP runtime for enough
# hardware queues BEFORE anything (torch included) initialises it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

FRAMES = 1000
N_ATOMS = 168
ALGO_BYTES_PER_UNIT = 24 * N_ATOMS + 600   # coordinates read once + one result record
ALGO_FLOP_PER_UNIT = 2.0e7
# the per-unit flop of section 8d by launch: evaluations x 168 atoms x 15 flop
ALGO_FLOP_BY_KERNEL = {
    "chains": (500 * N_ATOMS + N_ATOMS * (N_ATOMS + 1) // 2) * 15.0,            # F_opt evaluations + max_dim pairs
    "average": (947 * N_ATOMS + N_ATOMS * (N_ATOMS + 1) // 2) * 15.0,           # P_avg ray tests + max_dim pairs
    "windows": ((3100 + 2400 + 797) * N_ATOMS + N_ATOMS * (N_ATOMS + 1) // 2) * 15.0,   # paths, window fits, rays
}
# (round 6: the average diameter is a stage of the window launch's teams -- its flop belong to that launch)
ALGO_FLOP_BY_KERNEL["windows+average"] = ALGO_FLOP_BY_KERNEL["windows"] + ALGO_FLOP_BY_KERNEL["average"]
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6
# static inputs measured with rocprofv3 --pmc (separate passes; committed summaries), NOT by this run
TRAFFIC_FILES = ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01j_hbm_traffic.json")
COUNTER_FILES = ("r06_instruction_counters.json", "r05_instruction_counters.json", "r04_instruction_counters.json", "r03_instruction_counters.json", "r02_instruction_counters.json", "r01j_instruction_counters.json")
FP64_FILES = ("r06_fp64_counters.json",)
# the only figure for this metric the reference's repository holds: 715-frame CC3 trajectory,
# traj.analysis(ncpus=8) in 286.5 s (examples/Example7_AnalysingTrajectorySingleMol.ipynb:569-575)
REFERENCE_NOTEBOOK_FPS = 715 / 286.5
REFERENCE_SURVEY_FPS_PER_CORE = 0.54       # the reference itself, survey container, one core


# ---------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1).  They run BEFORE this process initialises the GPU: their worker
# processes are fresh interpreters, and nothing here may start one once HIP is up.
def _cpu_worker(args):
    """One worker process: analyse frames wid, wid + stride, ... until the budget is spent."""
    kind, wid, stride, budget_s, n_frames = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    sys.path.insert(0, ROOT)
    from pywindow_amd import element_data as E
    from pywindow_amd import synth

    elements, base = synth.load_cc3_base()
    ids = E.element_ids(elements)
    vdw, mass = np.ascontiguousarray(E.VDW[ids]), np.ascontiguousarray(E.MASS[ids])

    def frame(k):
        return synth.quantise_like_history(synth.noisy_frame(base, synth.SEED_BASE + k))

    from oracle import pw_oracle as O

    O.build()

    def run(xyz):
        O.full_analysis(xyz, vdw, mass)

    run(frame(wid))                      # warm-up (imports, first-call costs), untimed
    n = 0
    t0 = time.perf_counter()
    k = wid
    while k < n_frames and (n < 1 or time.perf_counter() - t0 < budget_s):
        run(frame(k))
        n += 1
        k += stride
    return n, time.perf_counter() - t0


def _cpu_rate(kind, workers, budget_s):
    import multiprocessing as mp

    jobs = [(kind, w, workers, budget_s, 100000) for w in range(workers)]
    # one thread per worker process: the BLAS / OpenMP pools of numpy, scipy and scikit-learn read these
    # at import, so they are set before the workers start
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    if workers == 1:
        res = [_cpu_worker(jobs[0])]
    else:
        with mp.get_context("spawn").Pool(workers) as pool:
            res = pool.map(_cpu_worker, jobs)
    frames = sum(r[0] for r in res)
    return frames, sum(r[0] / r[1] for r in res)      # workers run side by side: rates add


def _same_source_rate(threads, budget_s):
    """frames/s of the C ABI's explicit host path (device = -1) with `threads` host threads."""
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    elements, frames = synth.synthetic_units(max(32, 24 * threads))
    ids = E.element_ids(elements)
    batch = _lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids])
    ctx = _lib.Context(-1, host_threads=threads)
    ctx.analyse(_lib.Batch.uniform(frames[: max(2, threads)], E.VDW[ids], E.MASS[ids]))      # warm-up, untimed
    n = 0
    t0 = time.perf_counter()
    while n < 1 or time.perf_counter() - t0 < budget_s:
        out = ctx.analyse(batch)
        assert (out["n_windows"] == 4).all()
        n += len(frames)
    dt = time.perf_counter() - t0
    ctx.close()
    return n, n / dt


def cpu_baseline(budget_s=8.0):
    """The oracle (numpy/scipy/sklearn restatement of the reference's path, bit-identical to it on the
    golden inputs) timed on this host: one core and all cores; and the kernel SOURCE compiled for the
    host with a one-lane team (tests/hostsim), the "same source" CPU figure."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a container may see every CPU of the host and still be allowed only a share of them (cgroup CPU
    # quota): more worker processes than that only add overhead
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fa, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fb:
                q, per = float(fa.read()), float(fb.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            quota = None
    workers = min(cores, 256)
    if quota is not None:
        workers = max(1, min(workers, int(quota + 0.5)))
    n1, r1 = _cpu_rate("oracle", 1, budget_s)
    out = {"value": r1, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n1} frames of the same synthetic trajectory, oracle/pw_oracle.py, 1 process, {budget_s:.0f} s",
           "host_cpu_count": os.cpu_count(), "host_cores_usable": cores, "cgroup_cpu_quota": quota}
    try:
        na, ra = _cpu_rate("oracle", workers, budget_s)
        out["all_core"] = {"value": ra, "unit": "frames/s", "cores": workers, "kind": "port",
                           "sample": f"{na} frames, one oracle process per core ({workers} processes), {budget_s:.0f} s each"}
    except Exception as exc:  # pragma: no cover - depends on the host
        out["all_core"] = {"error": repr(exc)}
    try:
        ns, rs = _same_source_rate(1, budget_s / 2)
        nsa, rsa = _same_source_rate(workers, budget_s / 2)
        out["same_source"] = {"value": rs, "unit": "frames/s", "cores": 1, "kind": "port",
                              "what": "the product's own host path: pw_analysis_batch on a device = -1 context "
                                      "(pywindow_amd/csrc/pw_unit.hpp compiled by g++ for a one-lane team, pw_hostpath.cpp)",
                              "sample": f"{ns} frames", "all_core": {"value": rsa, "cores": workers, "sample": f"{nsa} frames"}}
    except Exception as exc:  # pragma: no cover
        out["same_source"] = {"error": repr(exc)}
    out["reference_itself"] = {"value": REFERENCE_SURVEY_FPS_PER_CORE, "unit": "frames/s", "cores": 1,
                               "where": "survey container (BASELINE.md section 2); the reference's Python cannot travel to the GPU box"}
    return out


# ---------------------------------------------------------------------------------------------------
def _median_ms(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts)), [round(t, 3) for t in ts]


def _median_legs(legs):
    """Median of every numeric entry of a list of DLPOLY.last_timings dicts (flags / counts: the last one)."""
    if not legs:
        return None
    out = {}
    for k, v in legs[-1].items():
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            out[k] = v
        elif k.endswith("_ms"):
            out[k] = round(float(np.median([l[k] for l in legs])), 4)
        else:
            out[k] = v
    return out


def _cold_first_call(path):
    """File name -> records in a FRESH process, milliseconds by leg: importing the package and loading the library,
    creating the device context (streams, probe, neighbour tables), opening + indexing the file, the first
    analysis_records call (first launch of every kernel).  None when the child fails."""
    import subprocess

    code = (
        "import time, json, sys\n"
        "t0 = time.perf_counter()\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import pywindow_amd as pw\n"
        "from pywindow_amd import _lib, engine\n"
        "_lib.load()\n"
        "t1 = time.perf_counter()\n"
        "ctx = engine.context(0)\n"
        "t2 = time.perf_counter()\n"
        f"traj = pw.DLPOLY({str(path)!r})\n"
        "t3 = time.perf_counter()\n"
        "recs = traj.analysis_records(forcefield='opls', swap_atoms={'he': 'H'})\n"
        "t4 = time.perf_counter()\n"
        "recs2 = traj.analysis_records(forcefield='opls', swap_atoms={'he': 'H'})\n"
        "t5 = time.perf_counter()\n"
        "print(json.dumps({'import_and_load_ms': 1e3 * (t1 - t0), 'context_ms': 1e3 * (t2 - t1), 'open_index_ms': 1e3 * (t3 - t2),"
        " 'first_analysis_ms': 1e3 * (t4 - t3), 'second_analysis_ms': 1e3 * (t5 - t4), 'open_plus_first_analysis_ms': 1e3 * (t4 - t2),"
        " 'units': int(len(recs)), 'ok': bool((recs['status'] == 0).all())}))\n"
    )
    try:
        proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        return json.loads(line[-1]) if line else {"error": (proc.stderr or proc.stdout)[-400:]}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def secondary(ctx, vdw, mass):
    """BASELINE.json's other shapes, reported beside the headline (not part of `value`), each the
    median of five repetitions after a warm-up."""
    import tempfile

    import pywindow_amd as pw
    from pywindow_amd import _lib, synth
    from pywindow_amd import rebuild as rb

    out = {}
    _, big = synth.synthetic_units(4000, first=50000)
    res = ctx.upload(_lib.Batch.uniform(big, vdw, mass))
    res.time_launches(2)                                   # (warm-up: the workspaces grow on the first launch of a larger batch)
    ms = res.time_launches(10)
    res.free()
    out["throughput_batch"] = {"units": 4000, "launches": 10, "ms_per_launch": ms, "units_per_s": 4000 / (ms * 1e-3)}
    # HISTORY file -> records: native parse, H2D, the launch, D2H (DLPOLY.analysis_records)
    with tempfile.TemporaryDirectory() as tmp:
        path = synth.write_synthetic_history(os.path.join(tmp, "HISTORY"), FRAMES)
        traj = pw.DLPOLY(path)

        legs = []

        def e2e():
            recs = traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"})
            legs.append(dict(traj.last_timings))
            return recs

        med, reps = _median_ms(e2e, reps=9, warm=2)      # (host-side legs: the box's other tenants show up in them)

        # ... and what comes before the first frame can be read: opening and indexing the file (the DLPOLY constructor:
        # mmap, the scan for "timestep" records, the atom keys of frame 0 -- the counterpart of _check_history /
        # _map_history, reference trajectory.py:647-689, 768-833), warm, and the whole thing in a FRESH process
        def open_only():
            t = pw.DLPOLY(path)
            n = t.no_of_frames
            del t
            return n

        open_med, open_reps = _median_ms(open_only, reps=9, warm=2)
        cold = _cold_first_call(path)
        out["e2e_history_to_records"] = {"frames": FRAMES, "ms": med, "frames_per_s": FRAMES / (med * 1e-3), "reps_ms": reps,
                                         "open_index_ms": open_med, "open_index_reps_ms": open_reps,
                                         "open_plus_analysis_ms": open_med + med,
                                         "cold_first_call_ms": cold,
                                         "includes": "tokenising the HISTORY text, H2D, all launches, D2H of the records; the "
                                                     "analysis is launched first and the reader feeds it (streamed batch).  "
                                                     "open_index_ms: the DLPOLY(path) constructor, warm; open_plus_analysis_ms: "
                                                     "file name to records, warm; cold_first_call_ms: the same in a fresh "
                                                     "process (library load, context creation with its tables, first launch)",
                                         "breakdown_ms": _median_legs(legs[2:]),
                                         "breakdown_note": "host-side legs, medians: the analysis runs asynchronously -- what the "
                                                           "host sees of it is wait_download (which contains the D2H copy)"}

        def e2e_dicts():
            traj.analysis_output = {}
            traj.analysis(forcefield="opls", swap_atoms={"he": "H"})
            return traj.analysis_output

        med, reps = _median_ms(e2e_dicts)
        out["e2e_history_to_dicts"] = {"frames": FRAMES, "ms": med, "frames_per_s": FRAMES / (med * 1e-3), "reps_ms": reps,
                                       "includes": "the same plus the reference's nested properties dict per frame "
                                                   "(DLPOLY.analysis -> analysis_output[frame]['0'])"}
    cell = os.path.join(ROOT, "tests", "golden", "rebuild.npz")
    if os.path.exists(cell):
        g = np.load(cell)
        el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
        frames = 256
        rng = np.random.default_rng(4)
        coords = xyz[None] + rng.normal(0.0, 0.02, size=(frames,) + xyz.shape)
        topo = rb.CellTopology(el)
        lats = np.repeat(lat[None], frames, axis=0)
        from pywindow_amd import element_data as E

        ids = E.element_ids(el)
        cc, ll, inv = rb.pack_frames(coords, lats)
        state = {}

        def run():
            res, n_mol = ctx.resident_from_cells(topo, E.VDW[ids], cc, ll, inv, True)
            res.launch()
            state["recs"] = res.download()
            state["n_mol"] = n_mol
            res.free()

        med, reps = _median_ms(run)
        n_mol, recs = state["n_mol"], state["recs"]
        # ... and the same from a HISTORY file: 1024 frames (8192 cages) parsed, re-assembled and analysed in
        # pieces (DLPOLY.modular_records, the Example-8 flow of BASELINE config 4)
        with tempfile.TemporaryDirectory() as tmp:
            pframes = 1024
            hpath = synth.write_history(os.path.join(tmp, "HISTORY_periodic"), el,
                                        (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(pframes)),
                                        cell=np.asarray(lat, float).T)
            ptraj = pw.DLPOLY(hpath)
            pstate = {}

            plegs = []

            def prun():
                pstate["r"] = ptraj.modular_records("all", rebuild=True)
                plegs.append(dict(ptraj.last_timings))

            pmed, preps = _median_ms(prun, reps=5)
            precs = pstate["r"][0]
            out["periodic_history_e2e"] = {
                "workload": "1024-frame periodic DL_POLY HISTORY (8 CC3 cages per cell), file to records",
                "frames": pframes, "cages": int(len(precs)), "ms": pmed, "reps_ms": preps,
                "frames_per_s": pframes / (pmed * 1e-3), "cages_per_s": len(precs) / (pmed * 1e-3),
                "includes": "tokenising the text, H2D, periodic re-assembly, analysis of every cage, D2H",
                "breakdown_ms": _median_legs(plegs[1:]),
                "breakdown_note": "host-side legs, medians, summed over the pieces: tokenise_wait = waiting for the reader "
                                  "thread (it decodes the next piece beside the device work), rebuild = H2D + re-assembly "
                                  "launch + hand-over, wait_download = analysis not yet finished + D2H",
                "all_status_zero": bool((precs["status"] == 0).all())}
        out["periodic_cell"] = {
            "workload": "cubic cell, 8 CC3 cages / 1344 atoms per frame (tests/data/system_periodic.pdb + 0.02 A noise)",
            "frames": frames, "cages": int(n_mol.sum()), "ms": med, "reps_ms": reps,
            "frames_per_s": frames / (med * 1e-3), "cages_per_s": float(n_mol.sum()) / (med * 1e-3),
            "includes": "H2D of the frames, rebuild launch, on-device hand-over, analysis launch, D2H of the records",
            "all_cages_have_windows": bool((recs["n_windows"] > 0).all())}
    return out


def secondary_multi(ctx, vdw, mass, dist, rank, world, local_rank, backend, tdev, barrier, make_gather, max_over_ranks, args):
    """BASELINE configs 4 and 5 on N > 1 ranks (every rank calls this: it ends in collectives).
    (i) config 5, the screen: each rank its ceil(units / N) units resident, ONE launch, the gather of the records
        inside the timed region (RCCL all_gather_into_tensor over xGMI on the device buffers; gloo in rehearsals).
    (ii) config 4, the periodic trajectory: ONE HISTORY file per node in /dev/shm, DLPOLY.analysis(modular=True,
        rebuild=True) sharded by frame, ragged gather of the records to rank 0 (trajectory.py: _analysis_modular).
    Replaces, on the reference's side, Trajectory._analysis_parallel + pool.get (trajectory.py:496-586)."""
    import tempfile

    import pywindow_amd as pw
    from pywindow_amd import _lib, synth

    out = {}
    rec_bytes = _lib.UNIT_OUT_DTYPE.itemsize
    # ---- (i) the screen ----
    total = int(args.multi_units)
    per = -(-total // world)
    # Errors of upload, launch and download are LOCAL to a rank (out of memory, a PwTimeoutError, ...): a rank that raised
    # while the others went on into the all-gather would leave them waiting for ever.  So every local step runs under
    # `locally`, and the ranks exchange "did anybody fail" before each collective -- all of them skip the block together.
    fails = []

    def locally(fn):
        try:
            return fn()
        except Exception as exc:  # noqa: BLE001
            fails.append(repr(exc))
            return None

    def anybody_failed():
        return max_over_ranks(1.0 if fails else 0.0) > 0.0

    t_gen = time.perf_counter()
    mine = locally(lambda: synth.screen_units(per, first=rank * per)[1])
    t_gen = time.perf_counter() - t_gen
    res = locally(lambda: ctx.upload(_lib.Batch.uniform(mine, vdw, mass)))
    gather = locally(lambda: make_gather(res, per))

    def launch_and_wait():
        res.launch()
        res.sync()          # (a time-out of the launch is reported here, on this rank, before anybody enters the gather)

    screen_ok = not anybody_failed()
    if screen_ok:
        locally(launch_and_wait)                              # warm-up: workspaces grow, the communicator sees the size
        screen_ok = not anybody_failed()
    if screen_ok:
        gather(); barrier(res)
        t0 = time.perf_counter()
        locally(launch_and_wait)
        screen_ok = not anybody_failed()
    if screen_ok:
        gather()
        barrier(res)
        el = max_over_ranks(time.perf_counter() - t0)
        t0 = time.perf_counter()
        gather()
        barrier(res)
        el_g = max_over_ranks(time.perf_counter() - t0)
        mine_recs = locally(res.download)
        screen_ok = not anybody_failed()
    if not screen_ok:
        out["config5_screen"] = {"skipped": "a rank failed in a local step (all ranks left the block together)",
                                 "errors_this_rank": fails}
        if res is not None:
            locally(res.free)
    else:
        allrec = gather.records()
        ok = None
        if rank == 0:
            ok = bool(len(allrec) == world * per and allrec[:per].tobytes() == mine_recs.tobytes()
                      and (allrec["status"] == 0).all() and (allrec["n_atoms"] == N_ATOMS).all())
        out["config5_screen"] = {
            "workload": "combinatorial screen, %d units of 168 atoms (BASELINE configs[4]: 5000 cages x 100 frames), "
                        "%d per rank resident, ONE launch per rank + the gather of the records" % (world * per, per),
            "units": world * per, "units_per_rank": per, "ms": 1e3 * el, "units_per_s": world * per / el,
            "gather_alone_ms": 1e3 * el_g, "gather_bytes_per_rank": per * rec_bytes,
            "gather_bytes_received_per_rank": world * per * rec_bytes, "gather_ok": ok, "backend": backend,
            "generate_s_rank0": round(t_gen, 2), "windows_eq_4_rank0": int((mine_recs["n_windows"] == 4).sum()),
            "includes": "launch of the three-kernel pipeline on every rank, its completion (a local wait: a rank's time-out must "
                        "not leave the others in the collective), all_gather of the fixed-size records (in the timed region), "
                        "barrier; max over ranks.  Not included: generating and uploading the coordinates"}
        res.free()
    del mine
    # ---- (ii) the periodic trajectory ----
    cell = os.path.join(ROOT, "tests", "golden", "rebuild.npz")
    frames = int(args.multi_frames)
    if os.path.exists(cell) and frames > 0:
        g = np.load(cell)
        el_, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
        base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        hdir = os.path.join(base, "pw_bench_%d_%d" % (os.getuid(), frames))
        hpath = os.path.join(hdir, "HISTORY_periodic")
        t_w = 0.0
        failed = 0.0
        if local_rank == 0 and not os.path.exists(hpath):
            try:
                os.makedirs(hdir, exist_ok=True)
                t_w = time.perf_counter()
                distinct = [xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(64)]
                tmp = hpath + ".tmp%d" % os.getpid()
                synth.write_history_cycled(tmp, el_, distinct, frames, cell=np.asarray(lat, float).T)
                os.replace(tmp, hpath)
                t_w = time.perf_counter() - t_w
            except Exception:  # noqa: BLE001  (no room in /dev/shm, ...: every rank must learn of it, none may wait)
                failed = 1.0
        dist.barrier()
        traj = None
        if not failed:
            try:
                traj = pw.DLPOLY(hpath)
            except Exception:  # noqa: BLE001
                failed = 1.0
        if max_over_ranks(failed) > 0.0:           # (one collective: all ranks skip the block together)
            out["config4_periodic"] = {"skipped": "the shared HISTORY file could not be written or opened on some rank", "path": hpath}
            return out
        traj.analysis(frames=list(range(min(64 * world, frames))), modular=True, rebuild=True, lazy=True)     # warm-up
        traj.analysis_output = {}
        traj._stores = []
        barrier_plain = lambda: (dist.barrier())
        barrier_plain()
        t0 = time.perf_counter()
        traj.analysis(frames="all", modular=True, rebuild=True, lazy=True)
        barrier_plain()
        el4 = max_over_ranks(time.perf_counter() - t0)
        legs = dict(traj.last_timings)
        n_units = None
        all_ok = None
        if rank == 0:
            store = traj.analysis_store
            n_units = int(len(store.records))
            all_ok = bool((store.records["status"] == 0).all() and n_units == 8 * frames)
        out["config4_periodic"] = {
            "workload": "%d-frame periodic DL_POLY HISTORY (8 CC3 cages / 1344 atoms per cell; 64 distinct frames cycled), "
                        "one file per node in %s, DLPOLY.analysis(modular=True, rebuild=True) sharded by frame (BASELINE "
                        "configs[3])" % (frames, base),
            "frames": frames, "frames_per_rank": -(-frames // world), "cages": n_units, "ms": 1e3 * el4,
            "frames_per_s": frames / el4, "cages_per_s": (8 * frames) / el4, "file_mb": round(os.path.getsize(hpath) / 1e6, 1),
            "write_s": round(t_w, 2), "gather_ok": all_ok, "backend": backend,
            "breakdown_ms_rank0": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in legs.items()},
            "expected_gather_bytes_per_rank": 8 * (-(-frames // world)) * rec_bytes,
            "includes": "every rank: tokenising its frames of the shared file, H2D, periodic re-assembly, analysis of its cages, "
                        "D2H; then the ragged gather of records and (frame, molecule) tags to rank 0; max over ranks"}
        # the file is a gigabyte of shared memory: it goes when every rank is done with it
        del traj
        dist.barrier()
        if local_rank == 0:
            try:
                os.remove(hpath)
                os.rmdir(hdir)
            except OSError:
                pass
    return out


def _load_profile(names):
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                return json.load(fh), name
        except (OSError, ValueError):
            continue
    return None, None


def _csrc_sha16():
    """Hash of pywindow_amd/csrc/* as tests/tools/provenance.py computes it (what a profile summary is tied to)."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "pywindow_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _provenance_of(name, data=None):
    """Where a committed profile summary comes from (commit, date, hash of the kernels' sources -- written into the
    summary by tests/tools/provenance.py at the end of a profile round) and whether the kernels have changed since:
    {"file", "head", "date", "csrc_sha16", "stale"}; stale is None for summaries older than the stamps."""
    info = None
    if isinstance(data, dict):
        info = data.get("provenance")
    if info is None and name is not None:
        try:
            with open(os.path.join(ROOT, "profiles", name + ".provenance.json")) as fh:
                info = json.load(fh)
        except (OSError, ValueError):
            info = None
    out = {"file": None if name is None else "profiles/" + name, "head": None, "date": None, "csrc_sha16": None, "stale": None}
    if info:
        out.update({k: info.get(k) for k in ("head", "date", "csrc_sha16", "dirty")})
        try:
            out["stale"] = bool(info.get("csrc_sha16") != _csrc_sha16())
        except OSError:
            out["stale"] = None
    return out


SERIAL_STATS_FILES = ("r06_serial_kernel_stats.csv", "r05_serial_kernel_stats.csv", "r04_serial_kernel_stats.csv", "r03_serial_kernel_stats.csv")
# which launch a kernel name of the stats file belongs to (template arguments: waves per team, stage mask)
_KERNEL_OF = (("pw_analyse_kernel<1, 37u>", "chains"), ("pw_analyse_kernel<4, 98u>", "average"),
              ("pw_analyse_kernel<4, 120u>", "windows"), ("pw_analyse_kernel<4, 122u>", "windows+average"))


def _serial_kernel_ms():
    """Average kernel durations (ms) of an analysis' launches from the committed rocprofv3 summary of one analysis at
    a time -> ({"chains": .., "windows+average": ..} -- rounds 1-5: "chains", "average", "windows" --, file name) or
    (None, None)."""
    import csv

    for name in SERIAL_STATS_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path, newline="") as fh:
                rows = list(csv.DictReader(fh))
        except OSError:
            continue
        out = {}
        for row in rows:
            kname = row.get("Name") or row.get("KernelName") or ""
            avg = row.get("AverageNs") or row.get("Average") or row.get("AverageNs ")
            for needle, launch in _KERNEL_OF:
                if needle in kname and avg:
                    out[launch] = float(avg) * 1e-6
        if out:
            return out, name
    return None, None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """``python bench.py --gpus N`` without torchrun: start the N ranks as a child process (this
    process has not touched the GPU and never will) and hand its output and exit code on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--multi-units", type=int, default=500000,
                    help="N > 1: units of the config-5 screen, all ranks together (BASELINE: 5000 cages x 100 frames)")
    ap.add_argument("--multi-frames", type=int, default=10000,
                    help="N > 1: frames of the config-4 periodic trajectory, all ranks together (BASELINE: 10k frames x 8 cages)")
    ap.add_argument("--no-strong", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))

    cpu = None
    if rank == 0 and world_env == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()              # before the GPU is initialised (see above)

    import torch

    # torch is plumbing here (device memory, streams, the process group): its intra-op pool -- one thread per hardware
    # thread of the host, 256 on the GPU box against a cgroup quota of 16 cores -- otherwise sits beside the reader's
    # sixteen threads (file -> records: decode leg 0.87 ms with the pool, 0.60 without; tests/tools/e2e_context.py)
    torch.set_num_threads(1)

    dist = None
    backend = os.environ.get("PW_BENCH_BACKEND", "nccl")       # "gloo": rehearsal of the multi-rank path
    device_index = local_rank
    if "PW_BENCH_DEVICE" in os.environ:                        # ... with every rank on one GPU
        device_index = int(os.environ["PW_BENCH_DEVICE"])
    world = 1
    if world_env > 1 or os.environ.get("PW_BENCH_FORCE_DIST") == "1":      # (forced: one-rank rehearsal of the RCCL path)
        import torch.distributed as dist

        torch.cuda.set_device(device_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
        world = dist.get_world_size()     # what the process group reports, not what the flag asked for
        rank = dist.get_rank()

    from pywindow_amd import _lib, engine, synth
    from pywindow_amd import element_data as E
    from pywindow_amd import trajectory as T

    elements, _ = synth.load_cc3_base()
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    ctx = engine.context(device_index)
    tdev = torch.device("cuda", device_index)
    rec_bytes = _lib.UNIT_OUT_DTYPE.itemsize

    # where every rank runs, as the process group and the runtime report it -- so that "did the N ranks get N
    # different GPUs" can be answered from the line alone
    def placement():
        try:
            props = torch.cuda.get_device_properties(device_index)
            bus = getattr(props, "pci_bus_id", None)
            pci = None if bus is None else "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), bus, getattr(props, "pci_device_id", 0))
            name = props.name
        except Exception as exc:  # noqa: BLE001
            pci, name = f"unknown ({exc!r})", None
        return {"rank": rank, "local_rank": local_rank, "device": device_index, "pci_bus_id": pci, "gpu": name,
                "visible_devices": torch.cuda.device_count(), "host": socket.gethostname(), "pid": os.getpid()}

    ranks_info = [placement()]
    if dist is not None:
        every = [None] * world
        dist.all_gather_object(every, ranks_info[0])
        ranks_info = every

    def barrier(res):
        res.sync()                     # the engine's own HIP streams
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    class StepGather:
        """The all-gather of one step's records, queued behind the launch without a host wait."""

        def __init__(self, res, per):
            self.res, self.per = res, per
            self.views = {}
            self.out = [torch.empty(world * per * rec_bytes, dtype=torch.uint8, device=tdev) for _ in range(2)] \
                if backend == "nccl" else None
            self.k = 0
            self.last_host = None

        def __call__(self):
            if backend == "nccl":
                stream = torch.cuda.current_stream(tdev).cuda_stream
                ptr = self.res.results_ready(stream)
                view = self.views.get(ptr)
                if view is None:
                    view = self.views[ptr] = T.device_bytes_tensor(ptr, self.per * rec_bytes, tdev)
                dist.all_gather_into_tensor(self.out[self.k & 1], view)
                self.res.results_release(stream)
                self.k += 1
            else:                                              # gloo rehearsal: host records
                self.last_host = T.gather_records(self.res.download(), self.per * world, rank, world, dist)

        def records(self):
            """Gathered records of the latest step (all ranks' blocks, rank order) on the host."""
            if backend == "nccl":
                raw = self.out[(self.k - 1) & 1].cpu().numpy()
                return raw.view(_lib.UNIT_OUT_DTYPE)
            return self.last_host

    per_rank = []           # seconds of the latest timed region, by rank

    def timed(res, steps, warmup, gather):
        for _ in range(warmup):
            res.launch()
            if gather:
                gather()
        barrier(res)
        t0 = time.perf_counter()
        for _ in range(steps):
            res.launch()
            if gather:
                gather()
        barrier(res)
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev if backend == "nccl" else "cpu")
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank[:] = [float(x.item()) for x in every]
            elapsed = max(per_rank)                 # the job takes as long as its slowest rank
        else:
            per_rank[:] = [elapsed]
        return elapsed

    # ---- weak scaling (headline): every rank its own `frames` frames --------------------------------
    _, frames = synth.synthetic_units(args.frames, first=rank * args.frames)
    res = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
    gather = StepGather(res, args.frames) if dist is not None else None
    if gather is not None:
        # set-up, like creating the process group: RCCL finishes building its channels during the first
        # collectives of a communicator (measured: the first ~20 all-gathers cost up to 6 % of a 25-step
        # run, after that the gather is free) -- so the communicator is exercised before the W warm-up
        # steps, on the buffers the steps use, without any analysis launch in between
        res.launch()
        for _ in range(20):
            gather()
        barrier(res)
    retried_after = None
    try:
        elapsed = timed(res, args.steps, args.warmup, gather)
    except _lib.PwTimeoutError as exc:
        # a launch that gave up waiting for another (PW_E_TIMEOUT, reported with its details): that measurement is void;
        # the K steps are timed once more, the repeat is counted (config.retries) and the line says so.  (Never seen on
        # the 1000-frame workload.)
        if dist is not None:
            raise
        retried_after = str(exc)
        _lib.load().pw_context_count_retry(ctx._h)
        elapsed = timed(res, args.steps, args.warmup, gather)
    weak_per_rank = list(per_rank)
    out = res.download()
    ok = bool((out["status"] == 0).all())
    gather_ok = None
    if gather is not None:
        allrec = gather.records()
        if rank == 0:
            mine = allrec[: args.frames]
            gather_ok = bool(len(allrec) == world * args.frames and mine.tobytes() == out.tobytes()
                             and (allrec["status"] == 0).all() and (allrec["n_atoms"] == N_ATOMS).all())

    def max_over_ranks(seconds):
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=tdev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- strong scaling: ONE trajectory of `frames` frames split over the ranks -----------------------
    strong = None
    if dist is not None and not args.no_strong:
        per = -(-args.frames // world)
        lo, hi = T.shard_range(args.frames, rank, world)
        # the same decision on EVERY rank (a short last block would leave its rank out of the collectives
        # the others enter): only trajectories that split evenly are timed
        if args.frames % world == 0:
            _, mine = synth.synthetic_units(hi - lo, first=lo)
            res_s = ctx.upload(_lib.Batch.uniform(mine, vdw, mass))
            el_s = timed(res_s, args.steps, args.warmup, StepGather(res_s, per))
            strong = {"frames_total": args.frames, "frames_per_gpu": per, "ms_per_step": 1e3 * el_s / args.steps,
                      "value": args.frames * args.steps / el_s, "unit": "frames/s", "scaling": "strong",
                      "includes_gather": True,
                      "ms_per_step_by_rank": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps}}
            res_s.free()
        else:
            strong = {"skipped": f"{args.frames} frames do not split evenly over {world} ranks"}

    # ---- BASELINE configs 4 and 5 across the ranks (N > 1 only; every rank takes part) ------------------
    multi = None
    if dist is not None and world > 1 and not args.no_secondary:
        res.sync()
        try:
            multi = secondary_multi(ctx, vdw, mass, dist, rank, world, local_rank, backend, tdev, barrier,
                                    lambda r, per: StepGather(r, per), max_over_ranks, args)
        except Exception as exc:  # noqa: BLE001  (every rank raises or none: the collectives are symmetric)
            multi = {"error": repr(exc)}

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        value = world * args.frames * args.steps / elapsed
        # kernel duration measured with HIP events on the launch stream
        k_ms = res.time_launches(max(3, args.steps))
        # one launch on its own (no overlap with a neighbour): the latency of a single batch
        lat = []
        for _ in range(5):
            res.sync()
            t1 = time.perf_counter()
            res.launch()
            res.sync()
            lat.append(1e3 * (time.perf_counter() - t1))
        single_ms = float(np.median(lat))
        units_per_s = args.frames / (k_ms * 1e-3)
        achieved_gbs = units_per_s * ALGO_BYTES_PER_UNIT / 1e9
        achieved_tf = units_per_s * ALGO_FLOP_PER_UNIT / 1e12
        tj, tname = _load_profile(TRAFFIC_FILES)
        traffic = None
        if tj is not None:
            try:
                traffic = tj["per_launch_bytes"] * (args.frames / tj["units_per_launch"])
            except (KeyError, TypeError, ZeroDivisionError):
                traffic = None
        # VALU issue utilisation from the profiled wave-level instruction count (committed summary) and
        # the kernel time measured in this run: instructions x 4 SIMD cycles / (SIMDs x clock x time)
        cj, cname = _load_profile(COUNTER_FILES)
        valu_issue = None
        if cj is not None:
            try:
                per_unit = cj["per_launch"]["SQ_INSTS_VALU"] / cj["units_per_launch"]
                vi = cj["valu_issue"]
                valu_issue = {"wave_instructions_per_launch": per_unit * args.frames,
                              "frac": per_unit * args.frames * vi["simd_cycles_per_wave_instruction"]
                                      / (vi["simds"] * vi["clock_ghz"] * 1e9 * k_ms * 1e-3),
                              "source": f"static: profiles/{cname} (rocprofv3 --pmc SQ_INSTS_VALU), not measured by this run",
                              "provenance": _provenance_of(cname, cj)}
            except (KeyError, TypeError, ZeroDivisionError):
                valu_issue = None
        # what the vector ALUs really execute in double precision (rocprofv3 --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64, a
        # committed summary like the two above): executed flop per launch and the fraction of the FP64 vector peak that is,
        # at the kernel time measured in this run -- beside `frac`, which prices the ALGORITHMIC 2e7 flop per unit
        fj, fname = _load_profile(FP64_FILES)
        executed = None
        if fj is not None:
            try:
                ex_launch = fj["executed_fp64_flop_per_unit"] * args.frames
                executed = {"executed_fp64_flop_per_launch": ex_launch,
                            "frac_executed": ex_launch / (k_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                            "executed_over_algorithmic": ex_launch / (ALGO_FLOP_PER_UNIT * args.frames),
                            "by_kernel": {k: {"executed_fp64_flop": v["executed_fp64_flop"] * args.frames / fj["units_per_launch"],
                                              "fp64_share_of_valu": v.get("fp64_share_of_valu")} for k, v in fj["kernels"].items()},
                            "what": "(2 FMA + MUL + ADD + TRANS) F64 wave instructions x 64 lanes (inactive lanes included): what the "
                                    "ALUs are occupied with in double precision -- pruned evaluations do not appear, address "
                                    "arithmetic, integer work and the single-precision screens are not counted",
                            "source": f"static: profiles/{fname} (rocprofv3 --pmc), not measured by this run",
                            "provenance": _provenance_of(fname, fj)}
            except (KeyError, TypeError, ZeroDivisionError):
                executed = None
        # per launch of the pipeline, each on its own (no other launch in flight): HIP events on the
        # stream the kernel runs on; profiles/r02_serial_kernel_stats.csv is rocprofv3's view of the same
        # Two clocks per launch: `ms` = rocprofv3's average kernel duration of ONE analysis at a time (committed summary,
        # static: first wave to last wave of the kernel -- the figure the fractions use); `ms_events_live` = HIP events
        # on the launch's own stream in THIS run, which also contain the time the launch sits behind its gate and waits
        # for SIMD slots.
        per_kernel = None
        try:
            st = res.stage_times()
            prof, prof_name = _serial_kernel_ms()
            per_kernel = []
            if "average" not in st:           # (the average diameter ran inside the window teams)
                st = {("windows+average" if k == "windows" else k): v for k, v in st.items()}
            for name, ms_live in st.items():
                fl = ALGO_FLOP_BY_KERNEL[name] * args.frames
                ms = prof.get(name) if (prof and args.frames == FRAMES) else None
                entry = {"kernel": name, "ms": ms if ms is not None else ms_live, "ms_events_live": ms_live,
                         "ms_source": (f"static: profiles/{prof_name} (rocprofv3 --kernel-trace --stats, one analysis at a time)"
                                       if ms is not None else "HIP events of this run (no committed profile for this shape)"),
                         "algorithmic_flop_per_launch": fl}
                entry["achieved_tflops"] = fl / (entry["ms"] * 1e-3) / 1e12
                entry["frac_fp64_valu"] = entry["achieved_tflops"] / FP64_VECTOR_PEAK_TFLOPS
                per_kernel.append(entry)
        except Exception:  # noqa: BLE001 - e.g. a context that runs single launches has no per-launch times
            per_kernel = None
        line = {
            "metric": "trajectory frames/sec full_analysis (pore+windows), CC3 1k-frame",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "vs_baseline_note": "BASELINE.md holds no published number for this metric (section 1: none), so there is nothing to "
                                "divide by.  The comparisons that mean something are in cpu_baseline / vs_cpu_baseline: the oracle "
                                "on one core and on every core this container may use, and the kernels' own source on those cores. "
                                "(The reference's repository holds one incidental timing, a 2018 notebook on unknown hardware: "
                                "715 frames in 286.5 s on 8 processes = %.2f frames/s, "
                                "examples/Example7_AnalysingTrajectorySingleMol.ipynb:569-575.)" % REFERENCE_NOTEBOOK_FPS,
            "config": {"workload": "CC3 1000-frame synthetic DL_POLY trajectory (BASELINE configs[1]), "
                                   "per-frame pore+windows, 168 atoms/frame",
                       "frames_per_gpu": args.frames, "stages": "all", "results_ok": ok, "retried_after": retried_after,
                       "retries": {"this_context": ctx.retries, "process": _lib.retries_total(),
                                   "what": "analyses repeated after PW_E_TIMEOUT (a launch gave up waiting for another); 0 on a healthy device"},
                       "successive_steps_overlap": bool(ctx.pipelined), "pipelined": bool(ctx.pipelined), "gate_timeouts": ctx.gate_timeouts,
                       "single_step_latency_ms": single_ms,
                       "windows_eq_4": int((out["n_windows"] == 4).sum()),
                       "ms_per_step_by_rank": {"min": 1e3 * min(weak_per_rank) / args.steps,
                                               "max": 1e3 * max(weak_per_rank) / args.steps},
                       "gather_in_timed_region": dist is not None, "gather_ok": gather_ok,
                       "backend": backend if dist is not None else None,
                       "world_size_reported_by_backend": world if dist is not None else None,
                       "world_size_env": world_env,
                       "ranks": ranks_info,
                       "distinct_gpus": len({(r["host"], r["pci_bus_id"]) for r in ranks_info})},
            "roofline": {"bound": "fp64_valu", "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS,
                         "algorithmic_flop_per_launch": ALGO_FLOP_PER_UNIT * args.frames,
                         "executed_fp64_flop_per_launch": None if executed is None else executed["executed_fp64_flop_per_launch"],
                         "frac_executed": None if executed is None else executed["frac_executed"],
                         "executed": executed,
                         "kernel": "pw_analyse_kernel x 2 (one analysis = optimiser chains | window search with the average-diameter stage, concurrent)",
                         "kernel_ms": k_ms,
                         "kernel_ms_note": "HIP events on the launch streams around back-to-back analyses / their number: the steady-state "
                                           "period (successive analyses overlap); config.single_step_latency_ms is one analysis on its own",
                         "frac_single_analysis": (args.frames * ALGO_FLOP_PER_UNIT / (single_ms * 1e-3) / 1e12) / FP64_VECTOR_PEAK_TFLOPS,
                         "per_kernel": per_kernel,
                         "hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": achieved_gbs / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": ALGO_BYTES_PER_UNIT * args.frames},
                         "traffic": traffic,
                         "traffic_note": None if tname is None else
                         f"static: fabric bytes per launch from rocprofv3 PMC (profiles/{tname}), not measured by this run; "
                         "includes Infinity-Cache hits on the re-used per-team workspaces",
                         "traffic_provenance": _provenance_of(tname, tj),
                         "valu_issue_measured": valu_issue,
                         "profile_inputs": {
                             "what": "the static inputs of this line (traffic, VALU issue, per_kernel[*].ms) and the tree they were "
                                     "measured on; stale = the sources under pywindow_amd/csrc have changed since",
                             "csrc_sha16_now": _csrc_sha16(),
                             "traffic": _provenance_of(tname, tj), "counters": _provenance_of(cname, cj),
                             "fp64_counters": _provenance_of(fname, fj),
                             "serial_kernel_stats": _provenance_of(prof_name if per_kernel else None)}},
        }
        if strong is not None:
            if "value" in strong:
                # against `world` GPUs each as fast as this one analysing the whole trajectory on its own
                strong["one_gpu_frames_per_s"] = units_per_s
                strong["speedup"] = strong["value"] / units_per_s
                strong["efficiency"] = strong["value"] / (world * units_per_s)
                # first class, beside the weak headline
                line["value_strong"] = strong["value"]
                line["efficiency_strong"] = strong["efficiency"]
            line["strong"] = strong
            line["config"]["scaling_note"] = (
                "`value` is WEAK scaling: every rank analyses its own %d frames per step (what a job with more trajectories "
                "than GPUs does).  BASELINE's metric names ONE 1k-frame trajectory on 1/2/4/8 GPUs, i.e. STRONG scaling: that "
                "is `value_strong` / `efficiency_strong` (`strong`: the same trajectory split into %d contiguous shards, the "
                "gather of the records included).  A shard of a few hundred frames is bound by the latency of its slowest "
                "optimiser chain, not by throughput (DESIGN.md section 5), so value_strong stays far below value." % (args.frames, world))
        if world == 1:
            # what strong scaling of the 1000-frame trajectory can reach: a rank's share analysed on this
            # GPU (ranks are independent; the gather is not in these numbers)
            model = {}
            for n in (2, 4, 8):
                if args.frames % n:
                    continue
                _, part = synth.synthetic_units(args.frames // n)
                r2 = ctx.upload(_lib.Batch.uniform(part, vdw, mass))
                try:
                    ms = r2.time_launches(10)
                except _lib.PwHipError as exc:
                    model[str(n)] = {"error": repr(exc)}
                    continue
                finally:
                    r2.free()
                model[str(n)] = {"frames_per_gpu": args.frames // n, "ms_per_step": ms,
                                 "predicted_frames_per_s": args.frames / (ms * 1e-3)}
            line["strong_scaling_model_1gpu"] = model
        if multi is not None:
            line["secondary_multi"] = multi
        if not args.no_secondary and world == 1:
            # (the headline above stands on its own: a secondary figure that fails is reported as such, not fatal)
            try:
                line["secondary"] = secondary(ctx, vdw, mass)
            except Exception as exc:  # noqa: BLE001
                line["secondary"] = {"error": repr(exc)}
            # the end-to-end figures, where the driver keeps them (it preserves `config` whole): what the headline leaves
            # out -- reading the file, H2D, D2H, the cold start -- and BASELINE's other shapes on this GPU
            sec = line["secondary"]

            def pick(*path):
                cur = sec
                for k in path:
                    if not isinstance(cur, dict) or k not in cur:
                        return None
                    cur = cur[k]
                return cur

            cold = pick("e2e_history_to_records", "cold_first_call_ms")
            line["config"]["end_to_end"] = {
                "what": "medians measured by this run after the headline (details under `secondary`): the 1000-frame HISTORY "
                        "file to records (tokenise + H2D + analysis + D2H), the same from the file NAME, a fresh process "
                        "(import, context, open, first analysis), the 1024-frame periodic HISTORY (8 cages per frame: "
                        "tokenise + re-assembly + analysis), and a 4000-unit resident batch",
                "e2e_history_to_records_ms": pick("e2e_history_to_records", "ms"),
                "e2e_history_to_records_frames_per_s": pick("e2e_history_to_records", "frames_per_s"),
                "e2e_open_plus_analysis_ms": pick("e2e_history_to_records", "open_plus_analysis_ms"),
                "e2e_history_to_dicts_ms": pick("e2e_history_to_dicts", "ms"),
                "cold_first_call_ms": None if not isinstance(cold, dict) or "error" in cold else {
                    "total": sum(cold[k] for k in ("import_and_load_ms", "context_ms", "open_index_ms", "first_analysis_ms")),
                    "context_ms": cold["context_ms"], "first_analysis_ms": cold["first_analysis_ms"]},
                "periodic_history_e2e": {"ms": pick("periodic_history_e2e", "ms"), "cages_per_s": pick("periodic_history_e2e", "cages_per_s"),
                                         "breakdown_ms": pick("periodic_history_e2e", "breakdown_ms")},
                "periodic_cell_cages_per_s": pick("periodic_cell", "cages_per_s"),
                "throughput_batch_units_per_s": pick("throughput_batch", "units_per_s")}
        if cpu is not None:
            line["cpu_baseline"] = cpu
            line["vs_cpu_baseline"] = {"one_core": value / cpu["value"],
                                       "all_core": (value / cpu["all_core"]["value"]) if "value" in cpu.get("all_core", {}) else None,
                                       "target": "north_star: >= 10x reference-CPU frames/s at 1 GPU"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
