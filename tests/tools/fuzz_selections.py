"""Random frame selections of one HISTORY through DLPOLY.analysis_records on the device (streamed natively, streamed by
chunks, uploaded in one piece -- whatever the selection makes it) against the host path.  GPU box."""
import pathlib, sys, tempfile
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import pywindow_amd as pw
from pywindow_amd import synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
with tempfile.TemporaryDirectory() as tmp:
    path = synth.write_synthetic_history(pathlib.Path(tmp) / "H", 700)
    traj = pw.DLPOLY(path)
    kw = dict(forcefield="opls", swap_atoms={"he": "H"})
    ref = pw.DLPOLY(path).analysis_records(device=-1, **kw)
    bad = 0
    for trial in range(24):
        kind = trial % 6
        if kind == 0: sel = list(range(int(rng.integers(0, 300)), int(rng.integers(400, 700))))          # a long run: native streamed read
        elif kind == 1: sel = sorted(rng.choice(700, size=int(rng.integers(260, 500)), replace=False).tolist())   # long, with gaps: chunked appends
        elif kind == 2: sel = rng.choice(700, size=int(rng.integers(1, 200)), replace=False).tolist()    # short, any order
        elif kind == 3: sel = int(rng.integers(0, 700))                                                   # one frame
        elif kind == 4: sel = (int(rng.integers(0, 100)), int(rng.integers(400, 700)))                    # a tuple = a range
        else: sel = rng.choice(700, size=int(rng.integers(300, 600)), replace=False).tolist()            # long, any order
        recs = traj.analysis_records(frames=sel, **kw)
        idx = traj._select(sel)
        same = recs.tobytes() == ref[idx].tobytes()
        bad += not same
        print(f"trial {trial} kind {kind} frames {len(idx)} streamed {traj.last_timings.get('streamed')} pieces {traj.last_timings.get('pieces')} identical {same}", flush=True)
    print("selections that differ:", bad)
    sys.exit(1 if bad else 0)
