"""Development container only: the REFERENCE (imported from /root/reference with the golden generator's rdkit stub) against
the host path (device = -1) on random molecules, every compared quantity bit for bit.   python tests/tools/reference_probe.py [seed] [molecules] [min atoms] [max atoms]"""
import sys, pathlib, time, warnings, logging
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import make_golden as MG
from pywindow_amd import _lib, engine
warnings.filterwarnings("ignore"); logging.disable(logging.CRITICAL)
pw = MG.load_reference()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
pool = np.array(["C", "H", "N", "O", "S", "F", "Cl"])
n_mol = int(sys.argv[2]) if len(sys.argv) > 2 else 24
N_LO = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N_HI = int(sys.argv[4]) if len(sys.argv) > 4 else 140
mols = []
for k in range(n_mol):
    n = int(rng.integers(N_LO, N_HI))
    kind = k % 3
    p = rng.normal(size=(n, 3))
    if kind == 0:
        p = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(4.0, 9.0) + rng.normal(scale=0.3, size=(n, 3))
    elif kind == 1:
        r = np.where(rng.random(n) < 0.5, rng.uniform(4.0, 6.0), rng.uniform(8.0, 10.0))
        p = p / np.linalg.norm(p, axis=1)[:, None] * r[:, None]
    else:
        t = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([np.cos(t) * 7.0, np.sin(t) * 7.0, rng.normal(scale=1.5, size=n)], axis=1) + rng.normal(scale=0.4, size=(n, 3))
    el = pool[rng.integers(0, int(rng.integers(1, len(pool) + 1)), size=n)]
    mols.append((el, np.round(p + rng.normal(scale=3.0, size=3), 6)))
host = engine.analyse(mols, stages=_lib.STAGE_ALL, device=-1)
same = 0; diffs = []
t0 = time.time()
for u, (el, xyz) in enumerate(mols):
    try:
        ms = pw.MolecularSystem.load_system({"elements": np.array(el), "coordinates": np.array(xyz)}, "probe")
        props = ms.system_to_molecule().full_analysis()
    except Exception as exc:
        # a pore radius <= 0 inverts opt_pore_diameter's box: SciPy raises, the engine reports PW_ST_NEGATIVE_PORE and
        # the Python layer raises the same ValueError (engine.raise_on_uncomputable) -- the same answer
        if isinstance(exc, ValueError) and "upper bound is less" in str(exc) and int(host[u]["status"]) & 1:
            same += 1; refused = globals().get("refused", 0) + 1; continue
        diffs.append((u, "reference raised " + repr(exc)[:80], int(host[u]["status"]))); continue
    r = host[u]; bad = []
    def eq(a, b): return np.array_equal(np.asarray(a, float), np.asarray(b, float))
    if not eq(props["maximum_diameter"]["diameter"], r["maxd"]): bad.append("maxd")
    if not eq(props["average_diameter"], r["avg_d"]): bad.append("avg_d")
    if not eq(props["pore_diameter"]["diameter"], r["pore_d"]): bad.append("pore_d")
    if not eq(props["pore_diameter_opt"]["diameter"], r["pore_opt_d"]): bad.append("pore_opt_d")
    if not eq(props["pore_diameter_opt"]["centre_of_mass"], r["pore_opt_c"]): bad.append("pore_opt_c")
    wd = props["windows"]["diameters"]
    nw = -1 if wd is None else len(wd)
    if nw != int(r["n_windows"]): bad.append(f"n_windows {nw} vs {int(r['n_windows'])}")
    elif nw > 0 and nw <= 16:
        if not eq(wd, r["win_d"][:nw]): bad.append("win_d")
        if not eq(np.asarray(props["windows"]["centre_of_mass"]).reshape(-1), np.asarray(r["win_c"]).reshape(-1)[:3 * nw]): bad.append("win_c")
    if bad: diffs.append((u, bad, len(el)))
    else: same += 1
print(f"{n_mol} molecules: identical {same} (of them refused by both, negative pore: {globals().get('refused', 0)}), different {len(diffs)} ({time.time() - t0:.0f} s of reference)")
for d in diffs[:10]: print("  ", d)
