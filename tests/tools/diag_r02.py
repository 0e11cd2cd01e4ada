"""Round-2 diagnostics (GPU box): where host time goes in context creation / first launches, and the
captured window angles against the fixture."""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402


def T(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print(f"{label:40s} {1e3 * (time.perf_counter() - t0):10.1f} ms", flush=True)
    return r


elements, frames = T("synthetic_units(64)", lambda: synth.synthetic_units(64))
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
T("load library", _lib.load)
ctx = T("Context(0) #1", lambda: _lib.Context(0))
for n in (1, 1, 12, 12, 64, 64):
    T(f"analyse {n} units", lambda n=n: ctx.analyse(_lib.Batch.uniform(frames[:n], vdw, mass)))
T("ctx.close()", ctx.close)
for k in range(3):
    c2 = T(f"Context(0) #{k + 2}", lambda: _lib.Context(0))
    T("  analyse 5", lambda: c2.analyse(_lib.Batch.uniform(frames[:5], vdw, mass)))
    T("  analyse 5 again", lambda: c2.analyse(_lib.Batch.uniform(frames[:5], vdw, mass)))
    T("  close", c2.close)

from _util import load_group, group_batch  # noqa: E402

g = load_group("static")
off, xyz, v, m = group_batch(g)
ctx = _lib.Context(0)
out, dbg = T("analyse_debug static", lambda: ctx.analyse_debug(_lib.Batch(off, xyz, v, m)))
cols = {c: i for i, c in enumerate(g["win_table_cols"])}
rows = g["win_table"][g["win_unit"] == 0]
for c, row in enumerate(rows):
    w = dbg[0]["win"][c]
    print("window", c, "gpu angles", repr(w[3]), repr(w[4]), "| fixture", repr(row[cols["angle_1"]]), repr(row[cols["angle_2"]]),
          "| new_z", repr(w[5]), repr(-row[cols["z_lb"]]), "| z_x", repr(w[7]), repr(row[cols["z_x"]]))
