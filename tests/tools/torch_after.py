import sys, pathlib, tempfile
sys.path.insert(0, ".")
mode = sys.argv[1]
import numpy as np
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
if mode in ("analyse", "both"):
    el, fr = synth.synthetic_units(2)
    ids = E.element_ids(el)
    _lib.Context(0).analyse(_lib.Batch.uniform(fr, E.VDW[ids], E.MASS[ids]))
if mode in ("history", "both"):
    import pywindow_amd as pw
    with tempfile.TemporaryDirectory() as t:
        p = synth.write_synthetic_history(pathlib.Path(t) / "H", 3)
        tr = pw.DLPOLY(p)
        tr.read_coordinates(0, 3)
import torch
torch.cuda.set_device(0)
print(mode, "torch ok", torch.cuda.device_count())
