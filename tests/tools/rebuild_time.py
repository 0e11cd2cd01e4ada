"""Time pw_discrete_molecules on replicated frames of the periodic CC3 cell (GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from test_rebuild import CASES  # noqa: E402

from pywindow_amd import rebuild as RB  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
system = CASES["cc3_cell_md0"][0]
topo = RB.CellTopology(system["elements"])
coords = np.array([system["coordinates"]] * frames)
lat = np.array([system["lattice"]] * frames)
for rebuild in (True, False):
    RB.discrete_molecules_frames(topo, coords[:4], lat[:4], rebuild)
    t = time.time()
    n_mol, off, src, img, xyz = RB.discrete_molecules_frames(topo, coords, lat, rebuild)
    dt = time.time() - t
    print(f"rebuild={rebuild}: {frames} frames x {topo.n} atoms in {dt * 1e3:.1f} ms wall (H2D+kernel+D2H) "
          f"-> {frames / dt:.0f} frames/s; molecules per frame {n_mol[0]}")
# latency of one frame on its own (kernel + copies)
for rebuild in (True, False):
    RB.discrete_molecules_frames(topo, coords[:1], lat[:1], rebuild)
    t = time.time()
    for _ in range(10):
        RB.discrete_molecules_frames(topo, coords[:1], lat[:1], rebuild)
    print(f"rebuild={rebuild}: one frame alone {(time.time() - t) * 100:.2f} ms")
