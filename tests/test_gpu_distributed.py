"""Two processes (gloo rendezvous, both on GPU 0) analysing one trajectory: frames shard by rank,
the records are gathered on rank 0 and must equal the single-process result -- for the plain
per-frame analysis and for the modular (periodic, rebuilt) one with its ragged gather."""
import os
import socket

import numpy as np
import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, periodic, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pywindow_amd as pw

    traj = pw.DLPOLY(path)
    if periodic:
        traj.analysis(modular=True, rebuild=True, forcefield="opls")
    else:
        traj.analysis(forcefield="opls", swap_atoms={"he": "H"})
    if rank == 0:
        out = {f: {m: (v["pore_diameter_opt"]["diameter"], tuple(np.sort(v["windows"]["diameters"])))
                   for m, v in mols.items()} for f, mols in traj.analysis_output.items()}
        q.put(out)
    else:
        q.put(len(traj.analysis_output))
    dist.destroy_process_group()


def _run(path, periodic):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(path), periodic, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]      # (spawned ranks import torch afresh: minutes on a cold box)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert [g for g in got if not isinstance(g, dict)] == [0]
    return [g for g in got if isinstance(g, dict)][0]


def test_two_ranks_equal_one(hip_ctx, tmp_path):
    import pywindow_amd as pw
    from pywindow_amd import synth

    path = synth.write_synthetic_history(tmp_path / "HISTORY", 7)
    both = _run(path, False)
    one = pw.DLPOLY(path)
    one.analysis(forcefield="opls", swap_atoms={"he": "H"}, distributed=False)
    assert sorted(both) == list(range(7))
    for f in range(7):
        v = one.analysis_output[f]["0"]
        assert both[f]["0"] == (v["pore_diameter_opt"]["diameter"], tuple(np.sort(v["windows"]["diameters"])))


def test_two_ranks_modular_rebuild(hip_ctx, tmp_path):
    import pywindow_amd as pw
    from test_rebuild import write_periodic_history

    _g, path = write_periodic_history(tmp_path)
    both = _run(path, True)
    one = pw.DLPOLY(path)
    one.analysis(modular=True, rebuild=True, forcefield="opls", distributed=False)
    assert sorted(both) == [0, 1]
    for f in (0, 1):
        assert sorted(both[f]) == sorted(one.analysis_output[f])
        for m, v in one.analysis_output[f].items():
            assert both[f][m] == (v["pore_diameter_opt"]["diameter"], tuple(np.sort(v["windows"]["diameters"])))


def _nccl_worker(port, path, q):
    """One rank, RCCL backend, on its own GPU: device binding and the device-to-device gather."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    dev = torch.cuda.device_count() - 1          # the last GPU of the box: not simply "0" when there are several
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
    import pywindow_amd as pw
    from pywindow_amd import _lib, engine, synth
    from pywindow_amd import element_data as E
    from pywindow_amd import trajectory as T

    assert engine.resolve_device() == dev        # what torch made current, not a hard-wired 0
    before = torch.cuda.current_device()
    traj = pw.DLPOLY(path)
    traj.analysis(forcefield="opls", swap_atoms={"he": "H"})     # nccl -> _run_and_gather_on_device
    assert torch.cuda.current_device() == before                 # the library gave the device back
    assert engine.context().device == dev
    got = {f: v["0"]["pore_diameter_opt"]["diameter"] for f, v in traj.analysis_output.items()}
    # the stream-ordered gather of a launch that is still in flight equals a synchronous download
    elements, frames = synth.synthetic_units(48)
    ids = E.element_ids(elements)
    res = engine.context().upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
    ok = True
    for _ in range(3):
        res.launch()
        recs = T.gather_records_device(res, 48, 0, 1, dist, torch.device("cuda", dev))
        ok = ok and recs.tobytes() == res.download().tobytes()
    res.free()
    q.put((got, ok, dev))
    dist.destroy_process_group()


def test_nccl_rank_binds_its_gpu_and_gathers_on_device(hip_ctx, tmp_path):
    import torch.multiprocessing as mp

    import pywindow_amd as pw
    from pywindow_amd import synth

    path = synth.write_synthetic_history(tmp_path / "HISTORY", 9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), str(path), q))
    p.start()
    got, ok, dev = q.get(timeout=900)
    p.join(timeout=300)
    assert p.exitcode == 0
    one = pw.DLPOLY(path)
    one.analysis(forcefield="opls", swap_atoms={"he": "H"}, distributed=False)
    assert got == {f: v["0"]["pore_diameter_opt"]["diameter"] for f, v in one.analysis_output.items()}
    assert ok


def _two_gpu_worker(rank, world, port, path, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import pywindow_amd as pw
    from pywindow_amd import engine

    traj = pw.DLPOLY(path)
    traj.analysis(forcefield="opls", swap_atoms={"he": "H"})
    q.put((rank, engine.context().device,
           {f: v["0"]["pore_diameter_opt"]["diameter"] for f, v in traj.analysis_output.items()}))
    dist.destroy_process_group()


def test_two_ranks_on_two_gpus_over_rccl(hip_ctx, tmp_path):
    """Two ranks on two DIFFERENT GPUs, RCCL gather over xGMI (skipped on a one-GPU box)."""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import pywindow_amd as pw
    from pywindow_amd import synth

    path = synth.write_synthetic_history(tmp_path / "HISTORY", 9)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, 2, port, str(path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {g[0]: g for g in (q.get(timeout=900) for _ in procs)}
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got[0][1] == 0 and got[1][1] == 1
    one = pw.DLPOLY(path)
    one.analysis(forcefield="opls", swap_atoms={"he": "H"}, distributed=False)
    assert got[0][2] == {f: v["0"]["pore_diameter_opt"]["diameter"] for f, v in one.analysis_output.items()}
    assert got[1][2] == {}


def _run_bench_child(extra_env, gpus=2, frames=240, multi=None):
    """``python bench.py --gpus N`` as a FRESH child process (it starts its own ranks through
    torch.distributed.run before touching a GPU); returns the parsed JSON line."""
    import json
    import os
    import subprocess
    import sys

    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--frames", str(frames), "--no-cpu-baseline"]
    # multi = (units, frames): BASELINE configs 5 and 4 across the ranks at reduced size (bench.py: secondary_multi)
    cmd += ["--no-secondary"] if multi is None else ["--multi-units", str(multi[0]), "--multi-frames", str(multi[1])]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def _check_multi_rank_line(line, world, frames):
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["gather_in_timed_region"] is True and cfg["gather_ok"] is True and cfg["results_ok"] is True
    assert cfg["frames_per_gpu"] == frames
    by_rank = cfg["ms_per_step_by_rank"]
    assert 0 < by_rank["min"] <= by_rank["max"] and abs(by_rank["max"] - line["ms_per_step"]) < 1e-6
    assert line["value"] > 0 and abs(line["value"] - world * frames / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    strong = line["strong"]
    assert strong["scaling"] == "strong" and strong["frames_total"] == frames and strong["frames_per_gpu"] == frames // world
    assert strong["includes_gather"] is True and 0 < strong["efficiency"] and strong["speedup"] == pytest.approx(
        strong["efficiency"] * world)


def _check_multi_blocks(line, world, units, frames):
    """The blocks the first real SCALE run has to carry for BASELINE configs 4 and 5 (round-4 review, item 5)."""
    multi = line["secondary_multi"]
    assert "error" not in multi, multi
    c5 = multi["config5_screen"]
    per = -(-units // world)
    assert c5["units"] == world * per and c5["units_per_rank"] == per and c5["gather_ok"] is True
    assert c5["units_per_s"] > 0 and c5["gather_bytes_per_rank"] == per * 696 and c5["gather_alone_ms"] > 0
    c4 = multi["config4_periodic"]
    assert c4["frames"] == frames and c4["cages"] == 8 * frames and c4["gather_ok"] is True
    assert c4["frames_per_s"] > 0 and c4["frames_per_rank"] == -(-frames // world)
    assert c4["expected_gather_bytes_per_rank"] == 8 * c4["frames_per_rank"] * 696
    assert c4["breakdown_ms_rank0"]["pieces"] >= 1


@pytest.mark.gpu
def test_bench_two_ranks_executes_on_hardware_gloo():
    """The multi-rank command itself, executed: two ranks started by ``bench.py --gpus 2`` on this box's
    one GPU with the gloo backend (the rehearsal of the RCCL run the driver makes on eight): the process
    group forms, every step ends with the gather inside the timed region, rank 0 checks the gathered
    records against its own and prints the line, a strong-scaling block included.  (Trajectory fan-out of
    the reference: trajectory.py:553-586.)"""
    line = _run_bench_child({"PW_BENCH_BACKEND": "gloo", "PW_BENCH_DEVICE": "0"}, multi=(1500, 96))
    _check_multi_rank_line(line, 2, 240)
    assert line["config"]["backend"] == "gloo"
    _check_multi_blocks(line, 2, 1500, 96)


@pytest.mark.gpu
def test_bench_eight_ranks_dry_run_gloo():
    """The driver's SCALE run is ``bench.py --gpus 8`` on an eight-GPU node; nothing here can measure it, so it is
    rehearsed: EIGHT ranks (gloo, all on this box's one GPU), three steps.  The line must say what the real run
    has to say -- the world size the process group reports, a gather that rank 0 verified, per-rank step times,
    the strong-scaling block, and where every rank ran (device ordinal, PCI bus id), so that "did RCCL see eight
    ranks on eight GPUs" is answerable from the JSON alone."""
    line = _run_bench_child({"PW_BENCH_BACKEND": "gloo", "PW_BENCH_DEVICE": "0"}, gpus=8, frames=64, multi=(1000, 64))
    _check_multi_rank_line(line, 8, 64)
    _check_multi_blocks(line, 8, 1000, 64)
    cfg = line["config"]
    assert cfg["world_size_reported_by_backend"] == 8 and cfg["world_size_env"] == 8
    assert [r["rank"] for r in cfg["ranks"]] == list(range(8)) and [r["local_rank"] for r in cfg["ranks"]] == list(range(8))
    assert all(r["device"] == 0 and r["pci_bus_id"] for r in cfg["ranks"])
    assert cfg["distinct_gpus"] == 1            # (the rehearsal's one GPU; eight on the driver's node)
    assert len({r["pid"] for r in cfg["ranks"]}) == 8


@pytest.mark.gpu
def test_bench_two_ranks_executes_on_hardware_rccl():
    """... and over RCCL, one rank per GPU, where the box has two."""
    from pywindow_amd import _lib

    if _lib.load().pw_device_count() < 2:
        pytest.skip("needs two GPUs")
    line = _run_bench_child({})
    _check_multi_rank_line(line, 2, 240)
    assert line["config"]["backend"] == "nccl"


@pytest.mark.gpu
def test_bench_uneven_split_does_not_hang():
    """A trajectory that does not split evenly over the ranks: every rank takes the same decision (the
    strong-scaling block is skipped) instead of some entering collectives the others never reach."""
    line = _run_bench_child({"PW_BENCH_BACKEND": "gloo", "PW_BENCH_DEVICE": "0"}, gpus=2, frames=125)
    assert line["n_gpus"] == 2 and "skipped" in line["strong"]
