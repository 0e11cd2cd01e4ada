import sys, time, tempfile, pathlib, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if mode != "plain":
    import torch
    torch.cuda.init(); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
    if mode == "torch1":
        torch.set_num_threads(1)
import pywindow_amd as pw
from pywindow_amd import synth
with tempfile.TemporaryDirectory() as tmp:
    path = synth.write_synthetic_history(pathlib.Path(tmp) / "H", 1000)
    traj = pw.DLPOLY(path)
    ts = []; legs = []
    for rep in range(14):
        t0 = time.perf_counter(); traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"}); ts.append(1e3 * (time.perf_counter() - t0)); legs.append(traj.last_timings["tokenise_ms"])
    print(mode, "median %.3f" % np.median(ts[3:]), "decode leg median %.3f" % np.median(legs[3:]), "threads", os.cpu_count(), len(os.sched_getaffinity(0)))
