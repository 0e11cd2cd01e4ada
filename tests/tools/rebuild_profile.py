"""Build a -DPW_RB_PROFILE variant of the library and print the phase times of the periodic
re-assembly (team 0, one line per frame it processes).  GPU box:
    python tests/tools/rebuild_profile.py --build   (container)
    python tests/tools/rebuild_profile.py [frames]  (GPU box)"""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
csrc = ROOT / "pywindow_amd" / "csrc"
so = ROOT / "tests" / "tools" / "libpw_rbprof.so"
if "--build" in sys.argv:
    hip = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c"]
    subprocess.check_call(hip + ["-DPW_RB_PROFILE", str(csrc / "pw_rebuild.hip"), "-o", "/tmp/rbprof_rebuild.o"])
    # (the other translation units as the product build left them)
    rest = [str(csrc / o) for o in ("pw_kernels.o", "pw_kernels_big.o", "pw_shape.o", "pw_history.o", "pw_hostpath.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", "/tmp/rbprof_rebuild.o", *rest, "-o", str(so)])
    sys.exit(0)
import numpy as np  # noqa: E402
from test_rebuild import CASES  # noqa: E402

from pywindow_amd import _lib  # noqa: E402
from pywindow_amd import rebuild as RB  # noqa: E402

_lib.LIB_PATH = so
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
system = CASES["cc3_cell_md0"][0]
topo = RB.CellTopology(system["elements"])
coords = np.array([system["coordinates"]] * frames)
lat = np.array([system["lattice"]] * frames)
for rebuild in (True, False):
    print("rebuild =", rebuild, flush=True)
    RB.discrete_molecules_frames(topo, coords, lat, rebuild)
