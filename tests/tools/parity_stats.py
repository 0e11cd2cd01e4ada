"""Parity distribution of the HIP engine against the live oracle on many frames
(run on the GPU box): python tests/tools/parity_stats.py [n_frames] [first]"""
import sys, pathlib, json
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
from multiprocessing import Pool

def oracle_one(a):
    from oracle import pw_oracle as O
    xyz, vdw, mass = a
    r = O.full_analysis(xyz, vdw, mass)
    return {k: r[k] for k in ("maxd", "avg_d", "pore_d", "pore_opt_d", "n_windows", "win_d", "pore_opt_c", "win_c")}

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    from pywindow_amd import _lib, synth
    from pywindow_amd import element_data as E
    from oracle import pw_oracle as O
    O.build()
    elements, frames = synth.synthetic_units(n, first=first)
    ids = E.element_ids(elements)
    vdw, mass = E.VDW[ids], E.MASS[ids]
    out = _lib.Context(0).analyse(_lib.Batch.uniform(frames, vdw, mass))
    with Pool(min(64, n)) as p:
        refs = p.map(oracle_one, [(frames[k], vdw, mass) for k in range(n)], chunksize=1)
    exact = {k: 0 for k in ("maxd", "avg_d", "pore_d", "pore_opt_d")}
    wrel = []; cabs = []; nwin_ok = 0; centre_exact = 0
    for k in range(n):
        r, o = refs[k], out[k]
        for key in exact: exact[key] += float(o[key]) == r[key]
        centre_exact += np.array_equal(o["pore_opt_c"], r["pore_opt_c"])
        nwin_ok += int(o["n_windows"]) == r["n_windows"]
        m = r["n_windows"]
        if m > 0 and int(o["n_windows"]) == m:
            p_ = np.argsort(o["win_d"][:m]); q = np.argsort(r["win_d"][:m])
            wrel += list(np.abs(o["win_d"][:m][p_] - r["win_d"][:m][q]) / np.abs(r["win_d"][:m][q]))
            cabs += list(np.max(np.abs(np.asarray(o["win_c"][:m])[p_] - r["win_c"][:m][q]), axis=1))
    wrel = np.array(wrel); cabs = np.array(cabs)
    print(json.dumps({"frames": n, "first_seed_offset": first, "bit_identical": {**exact, "pore_opt_centre": int(centre_exact)},
                      "window_count_equal": nwin_ok, "windows": len(wrel),
                      "window_diameter_rel_err": {"zero": int((wrel == 0).sum()), "le_1e-12": int((wrel <= 1e-12).sum()),
                                                  "le_1e-9": int((wrel <= 1e-9).sum()), "le_1e-6": int((wrel <= 1e-6).sum()), "max": float(wrel.max())},
                      "window_centre_abs_err_max": float(cabs.max())}))
