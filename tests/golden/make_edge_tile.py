"""Development container only (imports /root/reference through make_golden's rdkit stub): the five molecules of
tests/tools/reference_probe.py (seeds 32, 34, 35) on which the N x N call of the distance primitive is decided by an
entry of the BLAS's edge tile -- maximum_diameter of the molecule itself, or of the molecule shifted to its centre of
mass (the radius of the sampling sphere of find_average_diameter and find_windows), one ulp away under the order of the
body of the matrix.  Writes tests/golden/edge_tile.npz: the molecules and what the REFERENCE returns for them."""
import logging, pathlib, sys, warnings
import numpy as np
HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE)); sys.path.insert(0, str(HERE.parents[1]))
import make_golden as MG

warnings.filterwarnings("ignore"); logging.disable(logging.CRITICAL)
POOL = np.array(["C", "H", "N", "O", "S", "F", "Cl"])


def molecule(seed, index):                 # the generator of tests/tools/reference_probe.py, up to molecule `index`
    rng = np.random.default_rng(seed)
    mol = None
    for k in range(index + 1):
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
        el = POOL[rng.integers(0, int(rng.integers(1, len(POOL) + 1)), size=n)]
        mol = (el, np.round(p + rng.normal(scale=3.0, size=3), 6))
    return mol


pw = MG.load_reference()
out = {}
cases = [(32, 94), (32, 148), (32, 221), (34, 103), (35, 158)]
for m, (seed, index) in enumerate(cases):
    el, xyz = molecule(seed, index)
    ms = pw.MolecularSystem.load_system({"elements": np.array(el), "coordinates": np.array(xyz)}, "edge_tile")
    props = ms.system_to_molecule().full_analysis()
    md = props["maximum_diameter"]
    wd = props["windows"]["diameters"]
    out["m%d_elements" % m] = np.array(el); out["m%d_coordinates" % m] = np.array(xyz)
    out["m%d_maxd" % m] = md["diameter"]; out["m%d_maxd_atoms" % m] = np.array([md["atom_1"], md["atom_2"]])
    out["m%d_avg_d" % m] = props["average_diameter"]; out["m%d_pore_d" % m] = props["pore_diameter"]["diameter"]
    out["m%d_pore_opt_d" % m] = props["pore_diameter_opt"]["diameter"]
    out["m%d_pore_opt_c" % m] = np.array(props["pore_diameter_opt"]["centre_of_mass"])
    out["m%d_win_d" % m] = np.zeros(0) if wd is None else np.array(wd)
    out["m%d_win_c" % m] = np.zeros((0, 3)) if wd is None else np.array(props["windows"]["centre_of_mass"])
    print("seed %d molecule %d: %d atoms, maxd %.17g, %d windows" % (seed, index, len(el), md["diameter"], len(out["m%d_win_d" % m])))
out["count"] = len(cases)
np.savez(HERE / "edge_tile.npz", **out)
print("written", HERE / "edge_tile.npz")
