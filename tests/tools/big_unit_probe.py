"""Largest molecule a launch accepts (GPU box): k CC3 cages side by side as ONE unit."""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

el, frames = synth.synthetic_units(16)
ctx = _lib.Context(0)
for k in (6, 8, 10, 12, 14, 16):
    x = np.concatenate([frames[j] + np.array([30.0 * j, 0, 0]) for j in range(k)])
    ids = E.element_ids(list(el) * k)
    try:
        import time
        batch = _lib.Batch(np.array([0, len(x)]), x, E.VDW[ids], E.MASS[ids])
        ctx.analyse(batch)
        t0 = time.perf_counter()
        out = ctx.analyse(batch)[0]
        ms = 1e3 * (time.perf_counter() - t0)
        print(k * 168, "atoms: ok, status", int(out["status"]), "maxd", float(out["maxd"]), "windows", int(out["n_windows"]),
              f"| one unit, upload + analysis + download: {ms:.2f} ms")
    except _lib.PwHipError as exc:
        print(k * 168, "atoms:", str(exc)[:120])
