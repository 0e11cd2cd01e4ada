"""Sub-phase timers of the optimiser chains (-DPW_PROFILE -DPW_LB_FINE build of pw_kernels.hip): shader cycles per
unit and per call, chains' stages only.   python tests/tools/profile_chains.py --build [-DFOO ...] ; ... [n_units]"""
import ctypes, json, pathlib, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
csrc = ROOT / "pywindow_amd" / "csrc"
tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), "fine")
so = ROOT / "tests" / "tools" / f"libpw_prof_{tag}.so"
if "--build" in sys.argv:
    obj = f"/tmp/pwk_prof_{tag}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DPW_PROFILE",
                           "-DPW_LB_FINE", *[a for a in sys.argv if a.startswith("-D")], "-c", str(csrc / "pw_kernels.hip"), "-o", obj])
    rest = [str(csrc / o) for o in ("pw_kernels_big.o", "pw_rebuild.o", "pw_shape.o", "pw_history.o", "pw_hostpath.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", obj, *rest, "-o", str(so)])
    sys.exit(0)
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
_lib.LIB_PATH = so
L = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("-")]
n = int(args[0]) if args else 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
stages = _lib.STAGE_BASIC | _lib.STAGE_OPT
res.launch(stages); res.sync()
recs = res.download()
buf = (ctypes.c_ulonglong * 32)()
L.pw_debug_stage_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.pw_debug_stage_ticks(ctx._h, buf)   # reset
ms = res.time_launches(1, stages)
L.pw_debug_stage_ticks(ctx._h, buf)
t = np.array(list(buf), float) / 2    # two launches (warm-up + timed)
nit = float(recs["opt_nit"].mean()); nfg = float(recs["opt_nfev"].mean()) / 4
coarse = {0: "opt.step", 1: "opt.eval", 16: "lb.cauchy", 17: "lb.formk", 18: "lb.cmprlb", 19: "lb.subsm", 20: "lb.lnsrlb", 21: "lb.matupd", 22: "lb.formt"}
fine = {2: "bmv(all calls)", 3: "bmv.loads+lower sum", 4: "bmv.solve_ut", 5: "bmv.solve_un", 6: "bmv.upper sum+check+store", 7: "cauchy.head", 15: "cauchy.ddot",
        14: "cauchy.tail", 8: "formk.wn1+wn", 9: "formk.potrf1", 10: "formk.trtrs", 11: "formk.syrk+potrf2", 12: "subsm.trsv x2", 13: "cmprlb.tail",
        23: "lnsrlb.dcsrch", 24: "eval.wave_gap4", 25: "lnsrlb.fg(all of it)", 26: "lnsrlb.head", 27: "fg.pre(fd steps)", 28: "fg.post(gradient)",
        29: "subsm.head", 30: "subsm.tail"}
out = {"units": n, "kernel_ms": ms, "mean_nit": nit, "mean_fg_calls": nfg,
       "us_per_unit(100MHz)": {v: round(t[k] / 100.0 / n, 2) for k, v in coarse.items()},
       "kcycles_per_unit": {v: round(t[k] / 1000.0 / n, 1) for k, v in fine.items()},
       "cycles_per_iteration": {v: round(t[k] / n / nit) for k, v in fine.items()}}
print(json.dumps(out, indent=1))
