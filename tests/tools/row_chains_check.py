"""PW_ROW_CHAINS=1 (four optimiser chains per wavefront) against the default one-wave chains: identical records?
steady-state period?  GPU box.  usage: row_chains_check.py [frames] [iters]"""
import os
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
ref = None
for rows in ("0", "1"):
    os.environ["PW_ROW_CHAINS"] = rows
    ctx = _lib.Context(0)
    res = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
    ms = [res.time_launches(iters) for _ in range(3)]
    res.launch()
    out = res.download()
    lat = []
    import time
    for _ in range(5):
        res.sync(); t0 = time.perf_counter(); res.launch(); res.sync(); lat.append(1e3 * (time.perf_counter() - t0))
    if ref is None:
        ref = out.copy()
    same = out.tobytes() == ref.tobytes()
    diff = [k for k in out.dtype.names if not (out[k] == ref[k]).all()] if not same else []
    print(f"PW_ROW_CHAINS={rows}: ms/step {min(ms):.3f} (runs {[round(m, 3) for m in ms]}) single {sorted(lat)[2]:.3f} ms "
          f"-> {n / min(ms) * 1e3:.0f} frames/s | identical to one-wave chains {same} {diff} status0 {(out['status'] == 0).all()}", flush=True)
    res.free()
    ctx.close()
