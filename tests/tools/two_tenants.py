"""Two (or more) processes analysing on ONE device at the same time: the co-tenancy the pipeline's bounded waits exist
for.  Every tenant runs `iterations` launch + RAW download rounds (no repeat after PW_E_TIMEOUT) of a
256-unit and a 1000-unit batch in turn on its own context and compares the records with its first ones.

    python tests/tools/two_tenants.py [tenants=2] [iterations=200]          (GPU box)

Prints one JSON line: per tenant iterations, time-outs (with their texts), gate expiries, mismatches, whether the
context ran the pipeline, seconds; and the totals."""
import json
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]

if len(sys.argv) > 1 and sys.argv[1] == "--tenant":
    sys.path.insert(0, str(ROOT))
    import numpy as np

    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E

    rank, iters = int(sys.argv[2]), int(sys.argv[3])
    L = _lib.load()
    elements, frames = synth.synthetic_units(1000)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    ctx = _lib.Context(0)
    res = {n: ctx.upload(_lib.Batch.uniform(frames[:n], vdw, mass)) for n in (256, 1000)}
    first = {}
    st = {"tenant": rank, "iterations": 0, "timeouts": 0, "mismatches": 0, "texts": [], "pipelined": bool(ctx.pipelined),
          "slowest_ms": 0.0}
    t_start = time.perf_counter()
    for it in range(iters):
        n = 256 if it % 2 else 1000
        out = np.zeros(n, dtype=_lib.UNIT_OUT_DTYPE)
        t0 = time.perf_counter()
        res[n].launch()
        rc = L.pw_resident_download(ctx._h, res[n]._h, out.ctypes.data)
        st["slowest_ms"] = max(st["slowest_ms"], 1e3 * (time.perf_counter() - t0))
        if rc == _lib.E_TIMEOUT:
            st["timeouts"] += 1
            if len(st["texts"]) < 4:
                st["texts"].append(L.pw_last_error().decode(errors="replace"))
            continue
        _lib._check(rc, "pw_resident_download")
        if n not in first:
            first[n] = out.copy()
        elif out.tobytes() != first[n].tobytes():
            st["mismatches"] += 1
        st["iterations"] += 1
    st["gates"] = ctx.gate_timeouts
    st["seconds"] = round(time.perf_counter() - t_start, 2)
    st["status0"] = bool(all((first[n]["status"] == 0).all() for n in first))
    print("TENANT " + json.dumps(st), flush=True)
    sys.exit(0)

tenants = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
procs = [subprocess.Popen([sys.executable, __file__, "--tenant", str(r), str(iters)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
         for r in range(tenants)]
outs = []
for p in procs:
    try:
        so, se = p.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        p.kill()
        so, se = p.communicate()
    line = [ln for ln in so.splitlines() if ln.startswith("TENANT ")]
    outs.append(json.loads(line[-1][7:]) if line else {"error": (se or so)[-600:], "rc": p.returncode})
tot = {"tenants": tenants, "iterations_each": iters,
       "timeouts": sum(o.get("timeouts", 0) for o in outs), "mismatches": sum(o.get("mismatches", 0) for o in outs),
       "completed": sum(o.get("iterations", 0) for o in outs), "failed_tenants": sum(1 for o in outs if "error" in o),
       "per_tenant": outs}
print(json.dumps(tot))
