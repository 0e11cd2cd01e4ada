"""The optimiser-chain launch of the PRODUCT library on its own (stages: basic + optimised pore): its duration is the
slowest chain's -- what one analysis on its own cannot be faster than.   python tests/tools/chains_only.py [n_units]"""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np

from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
stages = _lib.STAGE_BASIC | _lib.STAGE_OPT
res.launch(stages)
res.sync()
recs = res.download()
ms = min(res.time_launches(1, stages) for _ in range(5))
print(f"product chains-only launch ({n} units): {ms:.3f} ms; mean nfev {recs['opt_nfev'].mean():.1f} nit {recs['opt_nit'].mean():.1f}"
      f" | slowest: nit {int(recs['opt_nit'].max())} nfev {int(recs['opt_nfev'].max())}")
