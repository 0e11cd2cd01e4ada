"""Launch one stage mask a few times (for rocprofv3 runs): python run_stage.py <stages> [units] [iters]"""
import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
st = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000; it = int(sys.argv[3]) if len(sys.argv) > 3 else 3
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
for _ in range(it):
    res.launch(st)
try:
    res.sync()
    print("done", st, n, it)
except _lib.PwTimeoutError as e:
    # a counter pass (rocprofv3 --pmc) runs ONE kernel at a time, in an order of its own: a residency gate dispatched
    # ahead of its optimiser launch waits its limit out and reports it; the launches complete all the same (the
    # consumers run after the chains) and their counters are what this run is for.
    print("done", st, n, it, "-- with a residency gate expired under the profiler's kernel serialisation:", e)
