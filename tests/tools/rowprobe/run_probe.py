#!/usr/bin/env python3
"""Row-packed optimiser chains (four per wavefront) against the product's one-wave chains: same results?
how long for N units?  Measurement tool for the GPU box."""
import ctypes
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from pywindow_amd import _lib, engine, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

n_units = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
elements, frames = synth.synthetic_units(n_units)
ids = E.element_ids(elements)
vdw, mass = np.ascontiguousarray(E.VDW[ids]), np.ascontiguousarray(E.MASS[ids])
ctx = engine.context(0)
batch = _lib.Batch.uniform(frames, vdw, mass)
res = ctx.upload(batch)
stages = _lib.STAGE_BASIC | _lib.STAGE_OPT
ms_prod = res.time_launches(10, stages)
res.launch(stages)
recs = res.download()
print(f"product chains-only launch ({n_units} units): {ms_prod:.3f} ms; mean nfev {recs['opt_nfev'].mean():.1f} nit {recs['opt_nit'].mean():.1f}")

L = ctypes.CDLL(str(pathlib.Path(__file__).resolve().parent / "librowprobe.so"))
rin = np.zeros(n_units, dtype=[("x0", np.float64, (3,)), ("r", np.float64)])
rin["x0"] = recs["com"]
rin["r"] = recs["pore_d"] / 2.0        # pore_d = g * 2: exact
rout_t = np.dtype([("x", np.float64, (3,)), ("f", np.float64), ("nit", np.int32), ("nfev", np.int32), ("task", np.int32), ("pad", np.int32)])
xyz = np.ascontiguousarray(frames).reshape(-1)
for defer in (0, 2, 3, 4):
    rout = np.zeros(n_units, dtype=rout_t)
    ms = ctypes.c_float(0)
    rc = L.row_probe_run(ctypes.c_long(n_units), ctypes.c_int(frames.shape[1]), xyz.ctypes.data_as(ctypes.c_void_p),
                         vdw.ctypes.data_as(ctypes.c_void_p), rin.ctypes.data_as(ctypes.c_void_p),
                         rout.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(defer), ctypes.c_int(10), ctypes.byref(ms))
    same = int((rout["x"] == recs["pore_opt_c"]).all(axis=1).sum())
    print(f"row-packed defer={defer}: rc {rc} {ms.value:.3f} ms; identical centres {same}/{n_units}; nit equal "
          f"{int((rout['nit'] == recs['opt_nit']).sum())}, nfev equal {int((rout['nfev'] == recs['opt_nfev']).sum())}")
