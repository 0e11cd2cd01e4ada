"""torch initialises the GPU FIRST, then the library: resident launches, a streamed batch, the trajectory driver."""
import os, pathlib, sys, tempfile, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import torch
torch.cuda.init(); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
if "--streams" in sys.argv:
    ss = [torch.cuda.Stream() for _ in range(8)]
    for s in ss:
        with torch.cuda.stream(s):
            x = x + 1
    torch.cuda.synchronize()
import pywindow_amd as pw
from pywindow_amd import _lib, engine, synth, trajectory
from pywindow_amd import element_data as E
ctx = engine.context(0)
L = _lib.load()
print("pipelined:", L.pw_context_pipelined(ctx._h), flush=True)
el, frames = synth.synthetic_units(300)
ids = E.element_ids(el)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
res.launch(); ref = res.download(); print("resident launch ok", (ref["n_windows"] == 4).all(), flush=True)
print("period", res.time_launches(10), flush=True)
res.free()
buf = ctx.pinned_array(frames.shape); buf[:] = frames
sres = ctx.stream_begin(len(frames), E.VDW[ids], E.MASS[ids])
sres.launch()
t0 = time.perf_counter()
for lo in range(0, 300, 75):
    sres.append(buf[lo:lo + 75])
try:
    got = sres.download(); print("streamed ok", got.tobytes() == ref.tobytes(), "%.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
except Exception as exc:
    print("streamed FAILED:", exc, "%.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
sres.free()
