"""HISTORY file -> records, 1000 frames: the streamed path (launch first, the reader feeds it) against the one-piece
path (decode, upload, launch), medians of 9 after a warm-up, with the host-side legs (GPU box)."""
import pathlib, sys, tempfile, time
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import pywindow_amd as pw
from pywindow_amd import synth, trajectory

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with tempfile.TemporaryDirectory() as tmp:
    path = synth.write_synthetic_history(pathlib.Path(tmp) / "HISTORY", n)
    traj = pw.DLPOLY(path)
    ref = None
    for label, smin, chunks in (("one piece", 10 ** 9, 8), ("streamed x8", 256, 8), ("streamed x4", 256, 4), ("streamed x16", 256, 16),
                                ("streamed x32", 256, 32)):
        trajectory.STREAM_MIN, trajectory.STREAM_CHUNKS = smin, chunks
        ts, legs = [], []
        for k in range(10):
            t0 = time.perf_counter()
            recs = traj.analysis_records(forcefield="opls", swap_atoms={"he": "H"})
            ts.append(1e3 * (time.perf_counter() - t0))
            legs.append(dict(traj.last_timings))
        ref = recs.tobytes() if ref is None else ref
        med = float(np.median(ts[1:]))
        print("   wait legs per rep:", [round(l["wait_download_ms"], 2) for l in legs[1:]], "upload:", [round(l["upload_ms"], 2) for l in legs[1:]])
        leg = {k: round(float(np.median([l[k] for l in legs[1:]])), 3) for k in ("tokenise_ms", "upload_ms", "launch_ms", "wait_download_ms")}
        print(f"{label}: median {med:.3f} ms ({n / med * 1e3:.0f} frames/s) reps {[round(t, 2) for t in ts[1:]]} legs {leg} "
              f"identical {recs.tobytes() == ref}", flush=True)
