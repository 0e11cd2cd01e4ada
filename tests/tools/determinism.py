"""Run-to-run determinism of a large batch (GPU box): the same resident batch analysed R times, every
record compared with the first run's.  usage: determinism.py [units] [runs]"""
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

if os.environ.get("PW_LIB"):
    _lib.LIB_PATH = pathlib.Path(os.environ["PW_LIB"])
units = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
elements, base = synth.load_cc3_base()
ids = E.element_ids(elements)
rng = np.random.default_rng(99)
coords = base[None] + rng.normal(0.0, 0.10, size=(units,) + base.shape)
ctx = _lib.Context(0)
off = np.arange(units + 1, dtype=np.int64) * len(base)         # (per-atom radii: also what older builds of the library read)
res = ctx.upload(_lib.Batch(off, coords.reshape(-1, 3), np.tile(E.VDW[ids], units), np.tile(E.MASS[ids], units)))
res.launch()
ref = res.download()
bad_total = 0
for r in range(runs):
    res.launch()
    out = res.download()
    if out.tobytes() != ref.tobytes():
        rows = [u for u in range(units) if out[u].tobytes() != ref[u].tobytes()]
        bad_total += len(rows)
        for u in rows[:4]:
            fields = [k for k in out.dtype.names if not np.array_equal(out[u][k], ref[u][k])]
            print(f"run {r}: unit {u} differs in {fields}; opt_nit {out[u]['opt_nit']} vs {ref[u]['opt_nit']}, "
                  f"pore_opt_d {out[u]['pore_opt_d']!r} vs {ref[u]['pore_opt_d']!r}", flush=True)
print(f"units {units} runs {runs}: mismatching records in total {bad_total}")
if "--oracle" in sys.argv:
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
    from oracle import pw_oracle as O

    for u in [int(a) for a in sys.argv[sys.argv.index("--oracle") + 1:]]:
        o = O.full_analysis(coords[u], E.VDW[ids], E.MASS[ids])
        print(f"oracle unit {u}: pore_opt_d {o['pore_opt_d']!r} | first GPU run {ref[u]['pore_opt_d']!r} nit {ref[u]['opt_nit']}")
