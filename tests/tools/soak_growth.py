"""Soak of the pipeline's hand-offs where its launch time-outs were seen: batches of 64 / 8192 /
1000 units and periodic pieces in turn, each on a FRESH context -- so every iteration is the first launch after the
workspaces grow -- with the raw download (no repeat after PW_E_TIMEOUT), the records compared with the first ones of
their kind.  On a time-out: the queues of every set, the gate counters and the error text are printed.

    python tests/tools/soak_growth.py [iterations=300] [seed=1]         (GPU box; ~1-2 minutes)

Prints one JSON line: iterations, time-outs, retries, gate time-outs, mismatches."""
import ctypes
import json
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np

from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
from pywindow_amd import rebuild as rb

ROOT = pathlib.Path(__file__).resolve().parents[2]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
L = _lib.load()

elements, frames_all = synth.synthetic_units(8192)
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
c_el, c_xyz, c_lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
c_ids = E.element_ids(c_el)
topo = rb.CellTopology(c_el)
cell_frames = 192
c_coords = c_xyz[None] + np.random.default_rng(4).normal(0.0, 0.02, size=(cell_frames,) + c_xyz.shape)
c_pack = rb.pack_frames(c_coords, np.repeat(c_lat[None], cell_frames, axis=0))

first = {}
stats = {"iterations": 0, "timeouts": 0, "mismatches": 0, "gate_timeouts": {"tail": 0, "head": 0, "residency": 0},
         "by_kind": {}, "incidents": []}


def raw_download(ctx, res):
    """launch + download WITHOUT the binding's repeat: a time-out is seen as what it is"""
    out = np.zeros(res.n_units, dtype=_lib.UNIT_OUT_DTYPE)
    for _ in range(3):
        res.launch()
        rc = L.pw_resident_download(ctx._h, res._h, out.ctypes.data)
        if rc != _lib.E_RETRY:
            break
    return rc, out


t_start = time.perf_counter()
kinds = ["u64", "u8192", "u1000", "cells"]
for it in range(iters):
    kind = kinds[it % 4] if it % 7 else kinds[int(rng.integers(0, 4))]
    ctx = _lib.Context(0)
    try:
        if kind == "cells":
            res, n_mol = ctx.resident_from_cells(topo, E.VDW[c_ids], *c_pack, True)
        else:
            n = int(kind[1:])
            res = ctx.upload(_lib.Batch.uniform(frames_all[:n], vdw, mass))
        reps = 3 if kind != "u8192" else 2
        for rep in range(reps):
            rc, out = raw_download(ctx, res)
            if rc == _lib.E_TIMEOUT:
                stats["timeouts"] += 1
                msg = L.pw_last_error().decode(errors="replace")
                inc = {"iteration": it, "kind": kind, "rep": rep, "error": msg, "queues": ctx.queue_state(),
                       "gates": ctx.gate_timeouts, "pipelined": ctx.pipelined}
                stats["incidents"].append(inc)
                print("TIMEOUT", json.dumps(inc), flush=True)
                continue
            _lib._check(rc, "pw_resident_download")
            key = kind
            if key not in first:
                first[key] = out.copy()
            elif out.tobytes() != first[key].tobytes():
                stats["mismatches"] += 1
                print("MISMATCH", it, kind, rep, flush=True)
        gt = ctx.gate_timeouts
        for k in gt:
            stats["gate_timeouts"][k] += gt[k]
        stats["by_kind"][kind] = stats["by_kind"].get(kind, 0) + 1
        res.free()
    finally:
        ctx.close() if hasattr(ctx, "close") else None
        del ctx
    stats["iterations"] += 1
stats["retries_total"] = _lib.retries_total()
stats["seconds"] = round(time.perf_counter() - t_start, 1)
stats["incidents"] = stats["incidents"][:8]
print(json.dumps(stats))
