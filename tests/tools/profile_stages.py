"""Build a -DPW_PROFILE variant of the library and print in-kernel stage shares."""
import ctypes, pathlib, subprocess, sys, json
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
csrc = ROOT / "pywindow_amd" / "csrc"
so = ROOT / "tests" / "tools" / "libpw_prof.so"
so.parent.mkdir(exist_ok=True)
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DPW_PROFILE", *[a for a in sys.argv if a.startswith("-D")], "-c", str(csrc / "pw_kernels.hip"), "-o", "/tmp/pwk_prof.o"])
    # (the other translation units as the product build left them)
    rest = [str(csrc / o) for o in ("pw_kernels_big.o", "pw_rebuild.o", "pw_shape.o", "pw_history.o", "pw_hostpath.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "/tmp/pwk_prof.o", *rest, "-o", str(so)])
    sys.exit(0)
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
_lib.LIB_PATH = so
L = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
res.launch(); res.sync()
buf = (ctypes.c_ulonglong * 32)()
L.pw_debug_stage_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.pw_debug_stage_ticks(ctx._h, buf)   # reset
iters = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1     # (> 1: the steady state, analyses overlapping)
ms = res.time_launches(iters)
L.pw_debug_stage_ticks(ctx._h, buf)
names2 = {14:"win.pre.shift",15:"win.pre.maxdim",16:"lb.cauchy",17:"lb.formk",18:"lb.cmprlb",19:"lb.subsm",20:"lb.lnsrlb",21:"lb.matupd",22:"lb.formt",24:"eps.knn",25:"eps.sum",26:"win.pre(shift,maxdim,points)",27:"avg.pre(shift,maxdim)",28:"avg.rays",29:"avg.compact+sum",11:"dbscan.adjacency",23:"dbscan.bfs",30:"smp.rays+compact",31:"smp.paths"}
names = ["opt.step", "consumer.wait", "win.path", "win.rotate", "win.z.step", "load_unit(all launches)", "win.brute", "win.nm", "eps", "sampling", "dbscan", "-", "windows(total)", "average"]
nl = iters + 1     # (pw_resident_time's warm-up launch counts too)
t = np.array(list(buf), float)[:14] / 100.0 / nl   # -> microseconds per launch
t2 = np.array(list(buf), float) / 100.0 / nl
d = {k: round(v / n, 2) for k, v in zip(names, t)}
d.update({v: round(t2[k] / n, 2) for k, v in names2.items()})
print(json.dumps({"units": n, "kernel_ms": ms, "us_per_unit": d}))
