"""End-to-end wall time of DLPOLY.analysis on a synthetic 1000-frame HISTORY (GPU box)."""
import pathlib
import sys
import tempfile
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import pywindow_amd as pw  # noqa: E402
from pywindow_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.perf_counter()
    path = synth.write_synthetic_history(pathlib.Path(tmp) / "HISTORY", n)
    t1 = time.perf_counter()
    traj = pw.DLPOLY(path)
    t2 = time.perf_counter()
    traj.analysis(forcefield="opls", swap_atoms={"he": "H"})          # includes context creation
    t3 = time.perf_counter()
    traj.analysis(forcefield="opls", swap_atoms={"he": "H"}, override=True)
    t4 = time.perf_counter()
    recs = traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"})
    t5 = time.perf_counter()
    traj.save_analysis(pathlib.Path(tmp) / "out.json")
    t6 = time.perf_counter()
print(f"frames {n}: write {t1-t0:.2f}s | open+index {1e3*(t2-t1):.1f} ms | first analysis {1e3*(t3-t2):.1f} ms | "
      f"second analysis (dicts) {1e3*(t4-t3):.1f} ms -> {n/(t4-t3):.0f} frames/s | columnar {1e3*(t5-t4):.1f} ms -> "
      f"{n/(t5-t4):.0f} frames/s | save json {1e3*(t6-t5):.1f} ms")
