"""Repeated launch + download of one resident batch of 2048 synthetic units: host wall time per repetition beside
the device-side stage times and the context's gate time-out counters.  On a busy pool single repetitions of
18-38 ms (against 4) were seen on some boxes; this is the tool that shows whether the time is on the device."""
import sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from pywindow_amd import engine, synth, _lib
from pywindow_amd import element_data as E
ctx = engine.context()
elements, frames = synth.synthetic_units(2048)
ids = E.element_ids(elements)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
ts = []
for rep in range(10):
    t = time.perf_counter(); res.launch(); t1 = time.perf_counter(); r = res.download(); t2 = time.perf_counter()
    st = res.stage_times()
    ts.append(1e3 * (t2 - t))
    print("rep %d: launch %.2f download %.2f ms; device stages %s" % (rep, 1e3 * (t1 - t), 1e3 * (t2 - t1), {k: round(v, 2) for k, v in st.items()}), flush=True)
print("A: pipelined", ctx.pipelined, "gate timeouts", ctx.gate_timeouts, "2048 synthetic units:", " ".join("%.2f" % t for t in ts), res.stage_times(), flush=True)
res.free()
