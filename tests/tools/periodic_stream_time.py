"""Long periodic trajectory through DLPOLY.modular_records (GPU box): one piece against several.  python tests/tools/periodic_stream_time.py [frames]"""
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
g = np.load(ROOT / "tests" / "golden" / "ptraj.npz")
rng = np.random.default_rng(3)
base = np.asarray(g["frames"][0], float)
with tempfile.TemporaryDirectory() as tmp:
    path = pathlib.Path(tmp) / "HISTORY"
    t0 = time.perf_counter()
    frames = (base + rng.normal(0.0, 0.01, size=base.shape) for _ in range(n))
    path.write_text(synth.history_text(g["elements"], frames, cell=g["cell"]))
    print(f"wrote {n} frames in {time.perf_counter() - t0:.1f} s", flush=True)
    traj = pw.DLPOLY(path)
    traj.modular_records(frames=list(range(8)), rebuild=True, forcefield="opls")      # warm-up
    for chunk in (10 ** 9, 1024):
        trajectory.MODULAR_CHUNK = chunk
        t0 = time.perf_counter()
        recs, uf, um = traj.modular_records(rebuild=True, forcefield="opls")
        dt = time.perf_counter() - t0
        print(f"chunk {chunk if chunk < 10 ** 9 else 'none'}: {n} frames, {len(recs)} cages in {dt * 1e3:.0f} ms -> "
              f"{n / dt:.0f} frames/s, {len(recs) / dt:.0f} cages/s; windows==4: {(recs['n_windows'] == 4).mean():.3f}",
              flush=True)
