"""HISTORY file -> records (DLPOLY.analysis_records), median of several repetitions (GPU box).
   python tests/tools/e2e_median.py [frames ...]"""
import pathlib
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import synth, trajectory  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1000, 10000]
with tempfile.TemporaryDirectory() as tmp:
    for n in sizes:
        path = synth.write_synthetic_history(pathlib.Path(tmp) / f"H{n}", n)
        traj = pw.DLPOLY(path)
        for label, piece in (("one piece", 10 ** 9), ("four pieces", max(1, n // 4))):
            trajectory.RUN_PIECE = piece if piece < 10 ** 9 else 10 ** 9
            ts = []
            for rep in range(7):
                t0 = time.perf_counter()
                recs = traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"})
                ts.append(1e3 * (time.perf_counter() - t0))
            print(f"frames {n} {label}: median {np.median(ts[2:]):.2f} ms ({n / np.median(ts[2:]) * 1e3:.0f} frames/s) "
                  f"reps {[round(t, 1) for t in ts]} status0 {(recs['status'] == 0).all()}", flush=True)
        trajectory.RUN_PIECE = 16384
