"""Timeline of the pieces of a periodic trajectory (GPU box): when each piece's tokenising, re-assembly call,
analysis launch and download start and end.  usage: periodic_pieces.py [frames]"""
import pathlib
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import _lib, synth, trajectory  # noqa: E402
from pywindow_amd import rebuild as rb  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = np.load(ROOT / "tests" / "golden" / "rebuild.npz")
el, xyz, lat = g["cc3_cell__in_elements"], g["cc3_cell__in_coordinates"], g["cc3_cell__in_lattice"]
log = []
t0 = [0.0]


def timed(name, fn):
    def wrapper(*a, **k):
        s = time.perf_counter()
        r = fn(*a, **k)
        log.append((name, (s - t0[0]) * 1e3, (time.perf_counter() - t0[0]) * 1e3))
        return r
    return wrapper


_lib.Context.resident_from_cells = timed("reassembly", _lib.Context.resident_from_cells)
_lib.Resident.download = timed("download", _lib.Resident.download)
_lib.Resident.launch = timed("launch", _lib.Resident.launch)
_lib.Resident.free = timed("free", _lib.Resident.free)
rb.pack_frames = timed("pack", rb.pack_frames)
trajectory.DLPOLY._read_selected = timed("tokenise", trajectory.DLPOLY._read_selected)
with tempfile.TemporaryDirectory() as tmp:
    path = pathlib.Path(tmp) / "H"
    synth.write_history(path, el, (xyz + np.random.default_rng(4 + k).normal(0.0, 0.02, size=xyz.shape) for k in range(n)),
                        cell=np.asarray(lat, float).T)
    for rep in range(3):
        traj = pw.DLPOLY(path)
        log.clear()
        t0[0] = time.perf_counter()
        traj.analysis(modular=True, rebuild=True)
        total = (time.perf_counter() - t0[0]) * 1e3
        print(f"rep {rep}: {n} frames in {total:.1f} ms = {n / total:.1f} k frames/s")
for name, a, b in sorted(log, key=lambda r: r[1]):
    print(f"{name:12s} {a:8.1f} -> {b:8.1f}  ({b - a:6.1f} ms)")
