"""Steady-state period of back-to-back 1000-frame analyses against the number of analyses in flight
(PW_SETS_IN_FLIGHT) and the gates (GPU box).  usage: sets_sweep.py [frames] [iters]"""
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

if os.environ.get("PW_LIB"):          # a variant build (tests/tools/build_variant.sh)
    _lib.LIB_PATH = pathlib.Path(os.environ["PW_LIB"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
ref = None
combos = [(2, 80, 85), (4, 60, 70), (4, 50, 50), (6, 50, 50), (8, 50, 50), (8, 30, 30), (8, 70, 70), (0, 50, 50)]
if len(sys.argv) > 3:
    combos = [tuple(int(x) for x in c.split(",")) for c in sys.argv[3:]]
for sets, tail, head in combos:
    os.environ["PW_SETS_IN_FLIGHT"] = str(sets)
    os.environ["PW_TAIL_GATE"] = str(tail)
    os.environ["PW_HEAD_GATE"] = str(head)
    ctx = _lib.Context(0)
    res = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
    ms = [res.time_launches(iters) for _ in range(3)]
    out = res.download()
    if ref is None:
        ref = out.tobytes()
    print(f"sets {sets} tail {tail} head {head}: ms/step {min(ms):.3f} (runs {[round(m, 3) for m in ms]}) "
          f"-> {n / min(ms) * 1e3:.0f} frames/s | identical {out.tobytes() == ref} status0 {(out['status'] == 0).all()}", flush=True)
    res.free()
    ctx.close()
