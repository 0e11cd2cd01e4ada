"""Two processes (gloo rendezvous, both on GPU 0) analysing one trajectory: frames shard by rank,
the records are gathered on rank 0 and must equal the single-process result -- for the plain
per-frame analysis and for the modular (periodic, rebuilt) one with its ragged gather."""
import os
import socket

import numpy as np
import pytest

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
