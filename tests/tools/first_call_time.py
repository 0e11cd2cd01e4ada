"""Where the first-call latency goes (GPU box): library load, context creation, first launch."""
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
t0 = time.perf_counter()
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

t1 = time.perf_counter()
_lib.load()
t2 = time.perf_counter()
ctx = _lib.Context(0)
t3 = time.perf_counter()
el, frames = synth.synthetic_units(4)
ids = E.element_ids(el)
batch = _lib.Batch.uniform(frames[:1], E.VDW[ids], E.MASS[ids])
t4 = time.perf_counter()
ctx.analyse(batch, _lib.STAGE_ALL)
t5 = time.perf_counter()
ctx.analyse(batch, _lib.STAGE_ALL)
t6 = time.perf_counter()
ctx.analyse(batch, _lib.STAGE_BASIC if hasattr(_lib, "STAGE_BASIC") else _lib.STAGE_ALL)
t7 = time.perf_counter()
print(f"import {1e3*(t1-t0):.1f} ms | dlopen {1e3*(t2-t1):.1f} ms | context {1e3*(t3-t2):.1f} ms | "
      f"first analyse(1 unit) {1e3*(t5-t4):.1f} ms | second {1e3*(t6-t5):.2f} ms | third {1e3*(t7-t6):.2f} ms")
