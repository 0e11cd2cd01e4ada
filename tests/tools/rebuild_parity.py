"""Periodic re-assembly on the GPU against the oracle (oracle/pw_rebuild.py, itself pinned on the
reference's outputs) on many perturbed cells: the CC3 test cell with its contents shifted by a
random vector (wrapped back into the cell, so every frame is cut by the faces differently and the
centres of mass move towards the boundaries), plus noise.  GPU box:
    python tests/tools/rebuild_parity.py [frames]"""
import json
import pathlib
import sys
from multiprocessing import Pool

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def oracle_one(system):
    from oracle import pw_rebuild as R

    mols = R.discrete_molecules(system, rebuild=R.create_supercell(system))
    return [(np.asarray(m["coordinates"]).tobytes(), list(m["elements"])) for m in mols]


if __name__ == "__main__":
    from test_rebuild import CASES

    from pywindow_amd import rebuild as RB

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    base = CASES["cc3_cell"][0]
    lattice = np.asarray(base["lattice"], float)
    inv = np.linalg.inv(lattice)
    rng = np.random.default_rng(77)
    systems = []
    for k in range(n):
        xyz = np.asarray(base["coordinates"], float)
        frac = (inv @ xyz.T).T + rng.uniform(0, 1, 3)
        frac -= np.floor(frac)                               # wrapped into <0, 1)
        xyz = (lattice @ frac.T).T + rng.normal(0.0, (0.0, 0.02, 0.05, 0.15)[k % 4], size=xyz.shape)
        s = dict(base)
        s["coordinates"] = xyz
        systems.append(s)
    with Pool(min(32, n)) as p:
        want = p.map(oracle_one, systems, chunksize=1)
    topo = RB.CellTopology(base["elements"])
    coords = np.array([s["coordinates"] for s in systems])
    lats = np.array([lattice] * n)
    n_mol, off, src, img, xyz = RB.discrete_molecules_frames(topo, coords, lats, True)
    bad = 0
    sizes = {}
    for k in range(n):
        got = RB.molecules_from_output(systems[k], int(n_mol[k]), off[k], src[k], xyz[k])
        ok = len(got) == len(want[k]) and all(
            np.asarray(g["coordinates"]).tobytes() == w[0] and list(g["elements"]) == w[1] for g, w in zip(got, want[k]))
        bad += 0 if ok else 1
        for g in got:
            sizes[len(g["elements"])] = sizes.get(len(g["elements"]), 0) + 1
    print(json.dumps({"frames": n, "frames_identical_to_oracle": n - bad, "molecules": int(n_mol.sum()),
                      "molecule_sizes": {str(a): b for a, b in sorted(sizes.items())}}))
