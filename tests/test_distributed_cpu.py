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
