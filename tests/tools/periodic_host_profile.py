"""Where the host time of a periodic piece goes (GPU box): parse, pack, re-assembly call, launch, download."""
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import _lib, engine, synth  # noqa: E402
from pywindow_amd import rebuild as rb  # noqa: E402
from pywindow_amd.element_data import VDW, element_ids  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
with tempfile.TemporaryDirectory() as tmp:
    path = pathlib.Path(tmp) / "H"
    synth.write_history(path, el, (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(n)),
                        cell=np.asarray(lat, float).T)
    traj = pw.DLPOLY(path)
    ids = element_ids(traj.elements())
    topo = rb.CellTopology(traj.elements())
    ctx = engine.context()
    print("context: pipelined =", ctx.pipelined, flush=True)
    for rep in range(3):
        t = [time.perf_counter()]
        coords, lattice = traj._read_selected(list(range(n)), True); t.append(time.perf_counter())
        coords, la, inv = rb.pack_frames(coords, lattice); t.append(time.perf_counter())
        res, n_mol = ctx.resident_from_cells(topo, VDW[ids], coords, la, inv, True); t.append(time.perf_counter())
        res.launch(); t.append(time.perf_counter())
        recs = res.download(); t.append(time.perf_counter())
        res.free(); t.append(time.perf_counter())
        names = ["parse", "pack_frames", "resident_from_cells", "launch", "download(wait)", "free"]
        print("gate timeouts", ctx.gate_timeouts, end=" | ")
        print(f"frames {n}: " + " | ".join(f"{k} {1e3 * (b - a):.1f}" for k, a, b in zip(names, t[:-1], t[1:])) + f" | total {1e3 * (t[-1] - t[0]):.1f} ms", flush=True)
