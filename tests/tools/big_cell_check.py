"""A cell too large for the visit bit sets in LDS (2x2x2 copies of the CC3 test cell, 10 752 atoms):
the GPU's stamp-array path against the host-compiled kernel source.  GPU box."""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from test_rebuild import CASES, run_hostsim  # noqa: E402

from pywindow_amd import rebuild as RB  # noqa: E402

base = CASES["cc3_cell"][0]
lat = np.asarray(base["lattice"], float)
xyz0 = np.asarray(base["coordinates"], float)
shifts = [(a, b, c) for a in range(2) for b in range(2) for c in range(2)]
xyz = np.concatenate([xyz0 + lat @ np.array(s, float) for s in shifts])
system = {"elements": np.concatenate([np.asarray(base["elements"])] * 8), "coordinates": xyz, "lattice": lat * 2.0}
topo = RB.CellTopology(system["elements"])
n_mol, off, src, img, out = RB.discrete_molecules_frames(topo, xyz[None], (lat * 2.0)[None], True)
got = RB.molecules_from_output(system, int(n_mol[0]), off[0], src[0], out[0])
want, status = run_hostsim(ROOT / "tests" / "hostsim", system, True, with_bits=False)
same = len(got) == len(want) and all(np.array_equal(g["coordinates"], w["coordinates"]) and list(g["elements"]) == list(w["elements"])
                                     for g, w in zip(got, want))
print(f"{len(xyz)} atoms: GPU {len(got)} molecules, host build {len(want)} (status {status}); identical: {same}")
