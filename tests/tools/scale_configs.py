"""BASELINE.json's large shapes on ONE GPU (GPU box): config 4 (10 000 frames x 8 cages = 80 000
units) and config 5 (5000 cages x 100 frames = 500 000 units, 2 GB of coordinates).  Units are
CC3 + N(0, 0.1 A) noise drawn in bulk (one generator per shape; the per-unit seeding rule of
SURVEY.md 8d would take minutes of host time and changes nothing for the kernels).

Checked, size-independent: a random sample of units re-analysed in a small batch gives
byte-identical records; every unit has status 0.  Prints one JSON object."""
import json
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, synth  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

elements, base = synth.load_cc3_base()
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
ctx = _lib.Context(0)
report = {}
for name, units in (("config4_80000_units", 80000), ("config5_500000_units", 500000)):
    if len(sys.argv) > 1 and name.split("_")[0] not in sys.argv[1:]:
        continue
    rng = np.random.default_rng(len(name))
    t0 = time.perf_counter()
    coords = np.empty((units,) + base.shape)
    for s in range(0, units, 20000):
        e = min(units, s + 20000)
        coords[s:e] = base[None] + rng.normal(0.0, 0.10, size=(e - s,) + base.shape)
    t_gen = time.perf_counter() - t0
    batch = _lib.Batch.uniform(coords, vdw, mass)
    t0 = time.perf_counter()
    res = ctx.upload(batch)
    res.sync()
    t_up = time.perf_counter() - t0
    res.launch(); res.sync()                       # warm-up
    t0 = time.perf_counter()
    res.launch(); res.sync()
    t_run = time.perf_counter() - t0
    t0 = time.perf_counter()
    recs = res.download()
    t_down = time.perf_counter() - t0
    res.free()
    pick = np.sort(rng.choice(units, 256, replace=False))
    small = ctx.analyse(_lib.Batch.uniform(coords[pick], vdw, mass))
    report[name] = {
        "units": units, "coordinate_bytes": int(coords.nbytes), "host_generation_s": t_gen, "upload_s": t_up,
        "launch_s": t_run, "download_s": t_down, "units_per_s": units / t_run,
        "units_per_s_with_transfers": units / (t_up + t_run + t_down),
        "status_all_zero": bool((recs["status"] == 0).all()),
        "windows_eq_4": int((recs["n_windows"] == 4).sum()),
        "sample_256_identical_in_small_batch": bool(small.tobytes() == recs[pick].tobytes()),
    }
    del coords, batch, recs
print(json.dumps(report))
