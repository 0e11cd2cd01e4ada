"""Periodic HISTORY file -> records of every cage (DLPOLY.modular_records(rebuild=True)), median of
several repetitions (GPU box).   python tests/tools/periodic_median.py [frames]"""
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import synth, trajectory  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
with tempfile.TemporaryDirectory() as tmp:
    path = pathlib.Path(tmp) / "HISTORY_periodic"
    synth.write_history(path, el, (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(n)),
                        cell=np.asarray(lat, float).T)
    traj = pw.DLPOLY(path)
    ref = None
    for label, piece, fl in (("one piece", 10 ** 9, 2), ("pieces of 512, 2 in flight", 512, 2), ("pieces of 512, 3 in flight", 512, 3),
                             ("pieces of 1024, 2 in flight", 1024, 2), ("pieces of 256, 2 in flight", 256, 2), ("pieces of 256, 3 in flight", 256, 3),
                             ("pieces of 128, 3 in flight", 128, 3)):
        trajectory.MODULAR_PIECE = piece
        trajectory.MODULAR_IN_FLIGHT = fl
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            recs, uf, um = traj.modular_records("all", rebuild=True)
            ts.append(1e3 * (time.perf_counter() - t0))
        if ref is None:
            ref = recs.tobytes()
        print(f"frames {n} {label}: median {np.median(ts[1:]):.1f} ms ({n / np.median(ts[1:]) * 1e3:.0f} frames/s, "
              f"{len(recs) / np.median(ts[1:]) * 1e3:.0f} cages/s) reps {[round(t, 1) for t in ts]} identical {recs.tobytes() == ref}", flush=True)
