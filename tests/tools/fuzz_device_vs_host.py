"""Random molecules through every stage on the device and on the host path (the same source for a one-lane team):
every field of every record must agree.   python tests/tools/fuzz_device_vs_host.py [molecules] [seed]   (PW_FUZZ_MAX_ATOMS=400 for larger ones)"""
import pathlib, sys, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from pywindow_amd import _lib, engine

n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
MAX_ATOMS = int(__import__("os").environ.get("PW_FUZZ_MAX_ATOMS", "300"))     # 300: the records of profiles/ were made with it
pool = np.array(["C", "H", "N", "O", "S", "F", "Cl", "Br", "P", "I"])
batch = []
for k in range(n_mol):
    n = int(rng.integers(1, MAX_ATOMS + 1))
    kind = int(rng.integers(0, 5))
    p = rng.normal(size=(n, 3))
    if kind == 0:
        p = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(3.0, 12.0) + rng.normal(scale=rng.uniform(0.0, 0.5), size=(n, 3))
    elif kind == 1:
        p = p * rng.uniform(0.5, 6.0)
    elif kind == 2:
        r = np.where(rng.random(n) < 0.5, rng.uniform(4.0, 7.0), rng.uniform(9.0, 12.0))
        p = p / np.linalg.norm(p, axis=1)[:, None] * r[:, None]
        p[: n // 2] += rng.normal(scale=1.5, size=3)
    elif kind == 3:          # a lattice fragment: many equal distances
        g = np.array([(x, y, z) for x in range(-3, 4) for y in range(-3, 4) for z in range(-3, 4)], dtype=np.float64) * 1.6
        g = g[(np.linalg.norm(g, axis=1) > 3.0)]
        p = g[rng.permutation(len(g))[: min(n, len(g))]]
    else:                    # a torus-like ring of beads
        t = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([np.cos(t) * 8.0, np.sin(t) * 8.0, rng.normal(scale=1.0, size=n)], axis=1) + rng.normal(scale=0.4, size=(n, 3))
    n = len(p)
    el = pool[rng.integers(0, int(rng.integers(1, len(pool) + 1)), size=n)]
    batch.append((el, p + rng.normal(scale=rng.choice([0.0, 5.0, 500.0, 2.0e4, 1.0e5]), size=3)))
# --knobs: one random set of find_windows / window_analysis / opt_pore_diameter keywords for the whole batch
prm = None
if "--knobs" in sys.argv:
    kw = dict(adjust_windows=float(rng.uniform(0.3, 2.2)), adjust_average=float(rng.uniform(0.3, 2.2)),
              increment=float(rng.uniform(0.5, 2.5)), pore_opt=bool(rng.random() < 0.8), increment2=float(rng.uniform(0.05, 0.4)),
              lb_z=bool(rng.random() < 0.5), z_second_mini=bool(rng.random() < 0.5))
    if rng.random() < 0.4:
        # (closed boxes only: with an open side the objective is unbounded and the optimiser walks away for its 15 000
        # iterations -- in the reference too)
        kw["opt_bounds"] = [(-float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 3.0))) for _ in range(3)]
        if "--open" in sys.argv:      # ... unless asked for: runaway centres, PW_ST_PATH_TOO_LONG, seconds per unit
            kw["opt_bounds"] = [(None if rng.random() < 0.3 else a, None if rng.random() < 0.3 else b) for a, b in kw["opt_bounds"]]
    if rng.random() < 0.4:
        kw["opt_start"] = [float(v) for v in rng.normal(scale=0.5, size=3)]
    if rng.random() < 0.3:
        kw["z_bounds"] = (-float(rng.uniform(0.5, 4.0)), float(rng.uniform(0.5, 4.0)))
    print("knobs:", kw)
    prm = _lib.Params(**kw)
    # (custom starts / bounds are in absolute coordinates: keep the molecules at the origin)
    batch = [(el, xyz - xyz.mean(axis=0)) for el, xyz in batch]
t0 = time.time(); host = engine.analyse(batch, stages=_lib.STAGE_ALL, device=-1, params=prm); t1 = time.time()
dev = engine.analyse(batch, stages=_lib.STAGE_ALL, device=0, params=prm); t2 = time.time()
bad = {}
for k in host.dtype.names:
    a, b = host[k], dev[k]
    for u in range(n_mol):
        same = np.array_equal(a[u], b[u], equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a[u], b[u])
        if not same:
            bad.setdefault(u, []).append(k)
print(f"{n_mol} molecules (seed {seed}): host {t1 - t0:.1f} s, device {t2 - t1:.2f} s; records that differ: {len(bad)}")
print("status histogram:", dict(zip(*np.unique(host["status"], return_counts=True))), "windows:", dict(zip(*np.unique(host["n_windows"], return_counts=True))))
for u in list(bad)[:8]:
    print("  unit", u, "atoms", len(batch[u][0]), "fields", bad[u], "opt_nit", host[u]["opt_nit"], dev[u]["opt_nit"])
sys.exit(1 if bad else 0)
