"""The only collective on the path -- gathering fixed-size result records to rank 0 --
exercised with world_size 2 on the gloo backend (CPU)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pywindow_amd import _lib
    from pywindow_amd.trajectory import gather_records, shard_range

    lo, hi = shard_range(n_total, rank, world)
    local = np.zeros(hi - lo, dtype=_lib.UNIT_OUT_DTYPE)
    local["n_atoms"] = np.arange(lo, hi)
    local["pore_d"] = np.arange(lo, hi) * 0.5
    local["win_d"][:, 3] = np.arange(lo, hi) + 0.25
    out = gather_records(local, n_total, rank, world, dist)
    if rank == 0:
        q.put((out["n_atoms"].tolist(), out["pore_d"].tolist(), out["win_d"][:, 3].tolist()))
    else:
        q.put(len(out))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_gather_records_world2(n_total):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    full = [g for g in got if isinstance(g, tuple)][0]
    assert full[0] == list(range(n_total))
    assert full[1] == [0.5 * i for i in range(n_total)]
    assert full[2] == [i + 0.25 for i in range(n_total)]
    assert [g for g in got if not isinstance(g, tuple)] == [0]


def _ragged_worker(rank, world, port, counts, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pywindow_amd import _lib
    from pywindow_amd.trajectory import gather_ragged

    first = sum(counts[:rank])
    local = np.zeros(counts[rank], dtype=_lib.UNIT_OUT_DTYPE)
    local["n_atoms"] = np.arange(first, first + counts[rank])
    tags = np.arange(2 * first, 2 * (first + counts[rank]), dtype=np.int64)
    out = gather_ragged(local, rank, world, dist)
    tout = gather_ragged(tags, rank, world, dist)
    q.put((rank, out["n_atoms"].tolist(), tout.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(9, 8), (0, 5), (3, 0)])
def test_gather_ragged_world2(counts):
    """Modular analysis: ranks own different numbers of molecules (frames split into a
    varying number of cages); rank 0 receives all records in rank order."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {g[0]: g for g in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    total = sum(counts)
    assert got[0][1] == list(range(total)) and got[0][2] == list(range(2 * total))
    assert got[1][1] == [] and got[1][2] == []


def _driver_worker(rank, world, port, path, modular, q):
    """``DLPOLY.analysis`` under a two-rank gloo group with the device work replaced by a stand-in that
    encodes (frame, molecule) in the records: what is tested is the host logic of the N > 1 path --
    which frames a rank takes, what it contributes to the gather, what rank 0 assembles."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pywindow_amd import _lib
    from pywindow_amd import trajectory as T

    seen = []

    def fake_run(self, frames, vdw, mass, device):
        seen.extend(frames)
        recs = np.zeros(len(frames), dtype=_lib.UNIT_OUT_DTYPE)
        recs["n_atoms"] = 168
        recs["pore_d"] = np.asarray(frames, float) + 0.5
        recs["n_windows"] = 2
        recs["win_d"][:, 0] = np.asarray(frames, float)
        recs["win_d"][:, 1] = rank
        return recs

    def fake_run_modular(self, frames, rebuild, el, device):
        seen.extend(frames)
        n_mol = [2 + (f % 3) for f in frames]                 # ragged: 2..4 molecules per frame
        uf = np.repeat(np.asarray(frames, np.int64), n_mol)
        um = np.concatenate([np.arange(k) for k in n_mol]).astype(np.int64) if frames else np.zeros(0, np.int64)
        recs = np.zeros(len(uf), dtype=_lib.UNIT_OUT_DTYPE)
        recs["n_atoms"] = 100 + um
        recs["pore_d"] = uf * 10.0 + um
        recs["n_windows"] = -1
        return recs, uf, um

    T.DLPOLY._run = fake_run
    T.DLPOLY._run_modular = fake_run_modular
    traj = T.DLPOLY(path)
    traj.analysis(modular=modular, rebuild=False)
    out = {f: {m: (p["no_of_atoms"], p["pore_diameter"]["diameter"]) for m, p in mols.items()}
           for f, mols in traj.analysis_output.items()}
    q.put((rank, sorted(seen), out))
    dist.destroy_process_group()


@pytest.mark.parametrize("modular", [False, True])
def test_trajectory_driver_shards_and_gathers_world2(tmp_path, modular):
    import torch.multiprocessing as mp

    from pywindow_amd import synth

    path = synth.write_synthetic_history(tmp_path / "HISTORY", 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, 2, port, str(path), modular, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {g[0]: g for g in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # contiguous blocks of ceil(7 / 2) = 4 frames: rank 0 analyses 0..3, rank 1 analyses 4..6
    assert got[0][1] == [0, 1, 2, 3] and got[1][1] == [4, 5, 6]
    assert got[1][2] == {}                                     # only rank 0 holds the result
    out = got[0][2]
    assert sorted(out) == list(range(7))
    if not modular:
        assert all(list(out[f]) == ["0"] and out[f]["0"] == (168, f + 0.5) for f in range(7))
    else:
        for f in range(7):
            assert sorted(out[f]) == list(range(2 + f % 3))
            assert all(out[f][m] == (100 + m, f * 10.0 + m) for m in out[f])


def _failing_rank_worker(rank, world, port, q):
    """DLPOLY.analysis on two ranks where rank 1's LOCAL step raises: both ranks must leave the analysis with an
    exception -- rank 1 its own, rank 0 PwRankError -- instead of rank 0 waiting in the gather for ever."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pywindow_amd as pw
    from pywindow_amd import synth, trajectory

    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        path = synth.write_synthetic_history(os.path.join(tmp, "HISTORY"), 6)
        traj = pw.DLPOLY(path)

        def run(frames, vdw, mass, device):                      # (stands in for upload / launch / download)
            if rank == 1:
                raise ValueError("rank 1 cannot")
            return np.zeros(len(frames), dtype=trajectory._lib.UNIT_OUT_DTYPE)

        traj._run = run
        try:
            traj.analysis(forcefield="opls", swap_atoms={"he": "H"})
            q.put((rank, "returned"))
        except trajectory.PwRankError:
            q.put((rank, "PwRankError"))
        except ValueError as exc:
            q.put((rank, "ValueError: %s" % exc))
    dist.destroy_process_group()


def test_a_rank_that_fails_locally_takes_the_others_out_of_the_gather():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == {0: "PwRankError", 1: "ValueError: rank 1 cannot"}
