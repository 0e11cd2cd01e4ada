"""Bounded waits of the pipeline: every wait inside a kernel is limited by time WITHOUT PROGRESS of
the other side, a stalled analysis is reported as PW_E_TIMEOUT within the limit, and two processes that share the one
device finish their analyses without a time-out (the co-tenancy the round-5 review asked to be tested)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_two_processes_on_one_device_finish_without_a_timeout():
    """Two tenants, each 400 launch + RAW download rounds (no repeat after PW_E_TIMEOUT) of a 256- and a 1000-unit batch
    on a context of its own, at the same time on device 0: every record byte-identical to the tenant's first ones, and
    at most 1 % of the rounds may report a time-out (measured on MI355X: none, 2.3 ms per round and tenant, the slowest
    round 5 ms; three tenants: none, slowest 37 ms)."""
    proc = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "two_tenants.py"), "2", "400"],
                          capture_output=True, text=True, timeout=900)
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert line, proc.stderr[-800:]
    got = json.loads(line[-1])
    print(json.dumps(got))
    assert got["failed_tenants"] == 0, got
    assert got["mismatches"] == 0, got
    assert got["timeouts"] <= 8, got                       # 1 % of 2 x 400
    assert got["completed"] >= 2 * 400 - 8
    for t in got["per_tenant"]:
        assert t["status0"] and t["pipelined"], t
        assert t["gates"]["residency"] == 0, t


_STALL = r"""
import sys, time, json
sys.path.insert(0, %(root)r)
import numpy as np
from pywindow_amd import _lib, synth
from pywindow_amd import element_data as E
elements, frames = synth.synthetic_units(300)
ids = E.element_ids(elements)
vdw, mass = E.VDW[ids], E.MASS[ids]
ctx = _lib.Context(0)
whole = ctx.upload(_lib.Batch.uniform(frames, vdw, mass))
whole.launch()
expect = whole.download()
buf = ctx.pinned_array(frames.shape)
buf[:] = frames
res = ctx.stream_begin(len(frames), vdw, mass)
res.launch()
res.append(buf[:100])                      # ... and the reader never delivers the rest
t0 = time.perf_counter()
err = None
try:
    res.sync()
except _lib.PwTimeoutError as exc:
    err = str(exc)
waited = time.perf_counter() - t0
res.free()
whole.launch()
again = whole.download()
print(json.dumps({"error": err, "waited_s": waited, "recovered": again.tobytes() == expect.tobytes(),
                  "retries": ctx.retries, "pipelined": bool(ctx.pipelined)}))
"""


def test_a_stalled_streamed_batch_is_a_timeout_within_the_limit():
    """A streamed batch whose host side stops appending: the launches that wait for coordinates give up after
    PW_STREAM_LIMIT_MS without an append (300 ms here, 5 s by default), the waiting call reports PW_E_TIMEOUT -- it names
    the wait -- within the limit plus the time of the call, and the context's next analysis is unaffected.  In a process of
    its own: the limits are read when the context is created."""
    env = dict(os.environ, PW_STREAM_LIMIT_MS="300")
    proc = subprocess.run([sys.executable, "-c", _STALL % {"root": str(ROOT)}], capture_output=True, text=True, timeout=300, env=env)
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert line, proc.stderr[-800:]
    got = json.loads(line[-1])
    assert got["pipelined"], got
    assert got["error"] is not None and "streamed batch" in got["error"], got
    assert 0.25 <= got["waited_s"] < 1.0, got
    assert got["recovered"], got
    assert got["retries"] == 0, got          # (reported, not silently repeated)
