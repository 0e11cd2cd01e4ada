"""BASELINE configs[3] on ONE GPU: the 10 000-frame periodic HISTORY (8 CC3 cages per cell; 64 distinct noisy frames
cycled, 1.08 GB of text in /dev/shm) from the file to the records, DLPOLY.modular_records(rebuild=True), under the
schedules of the re-assembly (PW_MODULAR_GROUP).   python tests/tools/periodic_10k.py [frames=10000] [reps=2] [groups=8,0]"""
import json
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
groups = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,0").split(",")]
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
with tempfile.TemporaryDirectory(dir=base) as tmp:
    path = pathlib.Path(tmp) / "HISTORY_periodic"
    t0 = time.perf_counter()
    distinct = [xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(64)]
    synth.write_history_cycled(path, el, distinct, n, cell=np.asarray(lat, float).T)
    print(json.dumps({"file_mb": round(path.stat().st_size / 1e6, 1), "write_s": round(time.perf_counter() - t0, 2)}), flush=True)
    traj = pw.DLPOLY(path)
    first = None
    for grp in groups:
        os.environ["PW_MODULAR_GROUP"] = str(grp)
        for rep in range(reps):
            t0 = time.perf_counter()
            recs, uf, um = traj.modular_records("all", rebuild=True)
            ms = 1e3 * (time.perf_counter() - t0)
            if first is None:
                first = recs.copy()
            print(json.dumps({"group": grp, "rep": rep, "frames": n, "ms": round(ms, 1), "cages": int(len(recs)),
                              "cages_per_s": round(len(recs) / ms * 1e3), "status0": bool((recs["status"] == 0).all()),
                              "same_as_first": bool(recs.tobytes() == first.tobytes()),
                              "cycle_ok": bool(recs[: 64 * 8].tobytes() == recs[64 * 8: 128 * 8].tobytes()) if n >= 128 else None,
                              "legs": {k: (round(v, 1) if isinstance(v, float) else v) for k, v in traj.last_timings.items()}}), flush=True)
