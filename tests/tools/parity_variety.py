"""Live parity on molecules other than CC3 (GPU box): every molecule of the static golden group (60 to
468 atoms, 2 to 6 windows or none, different element sets) with fresh noise, GPU against the oracle.
   python tests/tools/parity_variety.py [copies] [sigma]"""
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from multiprocessing import Pool  # noqa: E402


def oracle_one(a):
    from oracle import pw_oracle as O

    xyz, vdw, mass = a
    try:
        r = O.full_analysis(xyz, vdw, mass)
    except ValueError as exc:           # non-porous: scipy's inverted bounds
        return {"error": str(exc)}
    return {k: r[k] for k in ("maxd", "maxd_i", "maxd_j", "avg_d", "pore_d", "pore_opt_d", "n_windows", "win_d", "pore_opt_c", "win_c")}


if __name__ == "__main__":
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    from _util import load_group, molecules
    from oracle import pw_oracle as O
    from pywindow_amd import _lib, engine
    from pywindow_amd import element_data as E

    O.build()
    g = load_group("static")
    mols = molecules(g)
    rng = np.random.default_rng(2026)
    units, jobs, names = [], [], []
    for u, (el, xyz) in enumerate(mols):
        ids = E.element_ids(el)
        for c in range(copies):
            x = xyz + rng.normal(0.0, sigma, size=xyz.shape)
            units.append((el, x))
            jobs.append((x, E.VDW[ids], E.MASS[ids]))
            names.append(f"{g['names'][u]}#{c}")
    out = engine.analyse(units)
    with Pool(min(32, len(jobs))) as p:
        refs = p.map(oracle_one, jobs, chunksize=1)
    bad = []
    stats = {"units": len(units), "windows": 0, "none_windows": 0, "negative_pore": 0}
    for k, (r, o) in enumerate(zip(refs, out)):
        if "error" in r:
            stats["negative_pore"] += 1
            if not int(o["status"]) & _lib.ST_NEGATIVE_PORE:
                bad.append((names[k], "negative pore not flagged"))
            continue
        for key in ("maxd", "avg_d", "pore_d", "pore_opt_d"):
            if float(o[key]) != r[key]:
                bad.append((names[k], key, float(o[key]), r[key]))
        if (int(o["maxd_i"]), int(o["maxd_j"])) != (r["maxd_i"], r["maxd_j"]):
            bad.append((names[k], "maxd atoms"))
        if not np.array_equal(o["pore_opt_c"], r["pore_opt_c"]):
            bad.append((names[k], "pore_opt_c"))
        if int(o["n_windows"]) != r["n_windows"]:
            bad.append((names[k], "n_windows", int(o["n_windows"]), r["n_windows"]))
            continue
        m = r["n_windows"]
        if m < 0:
            stats["none_windows"] += 1
        elif m > 0:
            stats["windows"] += m
            if not (np.array_equal(o["win_d"][:m], r["win_d"][:m]) and np.array_equal(np.asarray(o["win_c"][:m]), r["win_c"][:m])):
                bad.append((names[k], "windows", np.max(np.abs(o["win_d"][:m] - r["win_d"][:m]))))
    stats["mismatches"] = len(bad)
    stats["n_atoms"] = sorted({len(e) for e, _ in units})
    print(json.dumps(stats))
    for b in bad[:20]:
        print("MISMATCH", b)
