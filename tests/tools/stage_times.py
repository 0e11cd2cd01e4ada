"""Per-stage kernel time on the GPU (HIP-event timed launches with stage masks)."""
import sys, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
out = {}
for name, st in (("basic", 1), ("basic+avg", 3), ("basic+opt", 5), ("basic+opt+windows", 13), ("all", 15)):
    out[name] = res.time_launches(5, st)
print(json.dumps({"units": n, "ms": out}))
