"""The 1024-frame periodic HISTORY file to records, a few times (for a rocprofv3 --kernel-trace timeline and the host-side
legs): python tests/tools/periodic_timeline.py [frames=1024] [reps=4]"""
import json
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
with tempfile.TemporaryDirectory() as tmp:
    path = pathlib.Path(tmp) / "H"
    synth.write_history(path, el, (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(n)),
                        cell=np.asarray(lat, float).T)
    traj = pw.DLPOLY(path)
    for rep in range(reps):
        t0 = time.perf_counter()
        recs, uf, um = traj.modular_records("all", rebuild=True)
        ms = 1e3 * (time.perf_counter() - t0)
        print(json.dumps({"rep": rep, "ms": round(ms, 2), "cages": int(len(recs)), "legs": {k: (round(v, 2) if isinstance(v, float) else v) for k, v in traj.last_timings.items()}}), flush=True)
