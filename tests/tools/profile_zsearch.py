"""The neck search of the window fits (Lbfgsb<1>, one per window) by routine: the optimiser's in-kernel timers of a
-DPW_ZPROF diagnostic build belong to the z-searches instead of the chains.

    python tests/tools/profile_stages.py --build -DPW_ZPROF                 (coarse: one timer per routine)
    python tests/tools/profile_chains.py --build --tag=zfine -DPW_ZPROF      (fine: sub-phases; window stage timers silent)
    python tests/tools/profile_zsearch.py [--fine] [units]                   (GPU box)"""
import ctypes, json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
fine_build = "--fine" in sys.argv
_lib.LIB_PATH = ROOT / "tests" / "tools" / ("libpw_prof_zfine.so" if fine_build else "libpw_prof.so")
L = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("-")]
n = int(args[0]) if args else 1000
elements, frames = synth.synthetic_units(n)
ids = E.element_ids(elements)
ctx = _lib.Context(0)
res = ctx.upload(_lib.Batch.uniform(frames, E.VDW[ids], E.MASS[ids]))
res.launch(); res.sync()
recs = res.download()
buf = (ctypes.c_ulonglong * 32)()
L.pw_debug_stage_ticks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.pw_debug_stage_ticks(ctx._h, buf)
res.time_launches(1)
L.pw_debug_stage_ticks(ctx._h, buf)
t = np.array(list(buf), float) / 2          # (pw_resident_time's warm-up launch + the timed one)
nw = float(recs["n_windows"].sum())
coarse = {16: "lb.cauchy", 17: "lb.formk", 18: "lb.cmprlb", 19: "lb.subsm", 20: "lb.lnsrlb", 21: "lb.matupd", 22: "lb.formt"}
out = {"units": n, "windows": nw, "us_per_window": {v: round(t[k] / 100.0 / nw, 2) for k, v in coarse.items()}}
if fine_build:
    fine = {2: "bmv(all calls)", 3: "bmv.loads+lower sum", 4: "bmv.solve_ut", 5: "bmv.solve_un", 6: "bmv.upper sum+check+store",
            7: "cauchy.head", 15: "cauchy.ddot", 14: "cauchy.tail", 8: "formk.wn1+wn", 9: "formk.potrf1", 10: "formk.trtrs",
            11: "formk.syrk+potrf2", 12: "subsm.trsv x2", 13: "cmprlb.tail", 23: "lnsrlb.dcsrch", 25: "lnsrlb.fg(all of it)",
            26: "lnsrlb.head", 29: "subsm.head", 30: "subsm.tail"}
    out["shader_cycles_per_window"] = {v: round(t[k] / nw) for k, v in fine.items()}
else:
    out["us_per_window"]["win.z.step (the whole search)"] = round(t[4] / 100.0 / nw, 2)
print(json.dumps(out))
