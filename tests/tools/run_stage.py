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
res.sync()
print("done", st, n, it)
