import sys, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
n = 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
out = ctx.analyse(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
for k in ("opt_nit", "opt_nfev", "n_eval", "n_survivors", "n_points"):
    v = out[k]
    print(k, "min", v.min(), "mean", round(float(v.mean()), 1), "p50", np.percentile(v, 50), "p90", np.percentile(v, 90), "p99", np.percentile(v, 99), "max", v.max())
print("tasks", np.unique(out["opt_task"], return_counts=True))
