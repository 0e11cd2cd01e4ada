"""What the waves of a window team wait at the team's barriers, by stage (-DPW_PROFILE -DPW_BARRIER_PROF build of
pw_kernels.hip): wave-microseconds per unit, summed over the team's four waves, beside the stage's elapsed time from
profile_stages.py.   python tests/tools/profile_barriers.py --build ; python tests/tools/profile_barriers.py [units]"""
import ctypes, json, pathlib, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
csrc = ROOT / "pywindow_amd" / "csrc"
so = ROOT / "tests" / "tools" / "libpw_prof_bar.so"
if "--build" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DPW_PROFILE",
                           "-DPW_BARRIER_PROF", "-c", str(csrc / "pw_kernels.hip"), "-o", "/tmp/pwk_prof_bar.o"])
    rest = [str(csrc / o) for o in ("pw_kernels_big.o", "pw_rebuild.o", "pw_shape.o", "pw_history.o", "pw_hostpath.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "/tmp/pwk_prof_bar.o", *rest, "-o", str(so)])
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
ms = res.time_launches(1)
L.pw_debug_stage_ticks(ctx._h, buf)
t = np.array(list(buf), float) / 100.0 / 2 / n      # wave-us per unit (two launches: warm-up + timed)
names = {5: "claim + load_unit", 27: "avg.pre", 28: "avg.rays", 29: "avg.compact+sum", 13: "average(rest)", 14: "win.pre.shift", 15: "win.pre.maxdim",
         26: "win.pre(rest: points)", 24: "eps.knn", 25: "eps.sum", 8: "eps(rest)", 30: "smp.rays+compact", 31: "smp.paths", 9: "sampling(rest)",
         10: "dbscan", 11: "dbscan.adjacency(inner)", 23: "dbscan.bfs(inner)", 12: "window fits (one cluster per wave)"}
out = {"units": n, "kernel_ms": ms, "barrier_wait_wave_us_per_unit": {v: round(float(t[k]), 2) for k, v in names.items()},
       "other_slots": {str(k): round(float(t[k]), 2) for k in range(32) if k not in names and t[k] > 0.005},
       "total_wave_us_per_unit": round(float(t.sum()), 1)}
print(json.dumps(out, indent=1))
