import sys, time, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pywindow_amd import _lib, engine, synth
el, xyz = synth.synthetic_units(6)
a = xyz[0].copy(); a[5, 1] = np.nan
b = xyz[1].copy(); b[7, 2] = np.inf
c = xyz[2].copy(); c[:] = 0.0
d = xyz[3].copy() * 1e150
e = xyz[4].copy(); e[0] = [1e308, -1e308, 1e308]
f = xyz[5].copy() * 1e-300
mols = [(el, a), (el, b), (el, c), (el, d), (el, e), (el, f), (el, xyz[0])]
dev = int(sys.argv[1])
t0 = time.time()
r = engine.analyse(mols, stages=_lib.STAGE_ALL, device=dev)
print("device", dev, "%.2f s" % (time.time() - t0), "status", list(r["status"]), "n_windows", list(r["n_windows"]), "nit", list(r["opt_nit"]))
