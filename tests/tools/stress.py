"""Robustness pass on the GPU box: odd batch sizes, repeated contexts, interleaved residents."""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

elements, frames = synth.synthetic_units(64)
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
ref = _lib.Context(0).analyse(_lib.Batch.uniform(frames, vdw, mass))
for n in (1, 2, 3, 5, 17, 64):
    for rep in range(2 if n == 3 else 1):     # (every context creates a dozen streams: keep their number modest)
        ctx = _lib.Context(0)
        out = ctx.analyse(_lib.Batch.uniform(frames[:n], vdw, mass))
        assert out.tobytes() == ref[:n].tobytes(), (n, rep)
        ctx.close()
ctx = _lib.Context(0)
a = ctx.upload(_lib.Batch.uniform(frames[:40], vdw, mass))
b = ctx.upload(_lib.Batch.uniform(frames[40:], vdw, mass))
for _ in range(5):          # two batches alternating on one context, overlapped launches
    a.launch(); b.launch(); a.launch(); b.launch()
ra, rb = a.download(), b.download()
assert ra.tobytes() == ref[:40].tobytes() and rb.tobytes() == ref[40:].tobytes()
a.launch(1); assert np.array_equal(a.download()["pore_d"], ref[:40]["pore_d"])
a.launch(); b.launch(2); a.launch()
assert a.download().tobytes() == ref[:40].tobytes()
print("stress ok")
