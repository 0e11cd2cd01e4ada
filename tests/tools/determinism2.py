"""Reproducer for a rare run-to-run difference: one unit's coordinates copied to every 4th position of
a large random batch; prints how many copies deviate from the majority result and their diagnostics."""
import os
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

if os.environ.get("PW_LIB"):
    _lib.LIB_PATH = pathlib.Path(os.environ["PW_LIB"])
units, which, runs = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 1577, int(sys.argv[2]) if len(sys.argv) > 2 else 5
stages = int(sys.argv[3]) if len(sys.argv) > 3 else 15
elements, base = synth.load_cc3_base()
ids = E.element_ids(elements)
rng = np.random.default_rng(99)
coords = base[None] + rng.normal(0.0, 0.10, size=(units,) + base.shape)
coords[0::4] = coords[which]
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(coords, E.VDW[ids], E.MASS[ids]))
for r in range(runs):
    res.launch(stages)
    out = res.download()
    c = out[0::4]
    vals, counts = np.unique(c["pore_opt_d"], return_counts=True)
    major = vals[np.argmax(counts)]
    bad = np.nonzero(c["pore_opt_d"] != major)[0]
    print(f"run {r}: copies {len(c)} majority pore_opt_d {major!r} deviating {len(bad)}", flush=True)
    for i in bad[:6]:
        print("   copy", i, "nit", c[i]["opt_nit"], "nfev", c[i]["opt_nfev"], "task", c[i]["opt_task"], "msg", c[i]["opt_msg"],
              "d", repr(c[i]["pore_opt_d"]), "| majority nit", c[np.argmax(c["pore_opt_d"] == major)]["opt_nit"], flush=True)
