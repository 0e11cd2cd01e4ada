"""Development container only (imports /root/reference through make_golden's rdkit stub): the molecule on which the
reference's L-BFGS-B takes a line-search step that ends ON a bound (stp == stpmx) -- SciPy's lnsrlb then puts the iterate
back on the bound ("take step and prevent rounding error beyond bound"); stp * d + t alone left it two ulps outside and
the run took twelve more objective evaluations.  Found with tests/tools/reference_probe.py (seed 23, molecule 116).
Writes tests/golden/bound_step.npz: the molecule and what the REFERENCE returns for it."""
import logging, pathlib, sys, warnings
import numpy as np
HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(HERE.parents[1]))
import make_golden as MG

warnings.filterwarnings("ignore"); logging.disable(logging.CRITICAL)
rng = np.random.default_rng(23)
pool = np.array(["C", "H", "N", "O", "S", "F", "Cl"])
mol = None
for k in range(117):                       # the generator of tests/tools/reference_probe.py, up to molecule 116
    n = int(rng.integers(20, 140)); kind = k % 3
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
    mol = (el, np.round(p + rng.normal(scale=3.0, size=3), 6))
el, xyz = mol
pw = MG.load_reference()
from pywindow._internal import utilities as U
ms = pw.MolecularSystem.load_system({"elements": np.array(el), "coordinates": np.array(xyz)}, "bound_step")
with MG.Capture(U, False) as cap:
    props = ms.system_to_molecule().full_analysis()
oc = [c for c in cap.minimize_calls if c["name"] == "correct_pore_diameter"][0]
np.savez(HERE / "bound_step.npz", elements=np.array(el), coordinates=np.array(xyz),
         maxd=props["maximum_diameter"]["diameter"], avg_d=props["average_diameter"], pore_d=props["pore_diameter"]["diameter"],
         pore_opt_d=props["pore_diameter_opt"]["diameter"], pore_opt_c=np.array(props["pore_diameter_opt"]["centre_of_mass"]),
         opt_nit=oc["nit"], opt_nfev=oc["nfev"], win_d=np.array(props["windows"]["diameters"]),
         win_c=np.array(props["windows"]["centre_of_mass"]))
print("written", HERE / "bound_step.npz", "nit", oc["nit"], "nfev", oc["nfev"])
