"""Host logic of the one-process-per-GPU path, on the CPU: which HIP device an analysis binds to, and
how ``bench.py --gpus N`` turns itself into N ranks (reference fan-out: Trajectory._analysis_parallel,
trajectory.py:553-586)."""
import importlib.util
import sys
import types

import pytest

from _util import ROOT
from pywindow_amd import engine


class _FakeCuda:
    def __init__(self, current, initialized=True, available=True):
        self._c, self._i, self._a = current, initialized, available

    def is_available(self):
        return self._a

    def is_initialized(self):
        return self._i

    def current_device(self):
        return self._c


def _fake_torch(monkeypatch, **kw):
    mod = types.ModuleType("torch")
    mod.cuda = _FakeCuda(**kw)
    monkeypatch.setitem(sys.modules, "torch", mod)


def test_device_resolution_order(monkeypatch):
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.delitem(sys.modules, "torch", raising=False)
    engine.set_default_device(None)
    assert engine.resolve_device() == 0                      # nothing said: device 0
    assert engine.resolve_device(3) == 3                     # explicit argument wins
    # torchrun: LOCAL_RANK (modulo the visible devices; no device here -> taken as is)
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(engine._lib.load(), "pw_device_count", lambda: 8, raising=False)
    assert engine.resolve_device() == 5
    monkeypatch.setattr(engine._lib.load(), "pw_device_count", lambda: 4, raising=False)
    assert engine.resolve_device() == 1
    # the application has made a GPU current in PyTorch (torch.cuda.set_device(local_rank)): that one
    _fake_torch(monkeypatch, current=6)
    assert engine.resolve_device() == 6
    # ... but an imported torch that never touched its GPU runtime says nothing
    _fake_torch(monkeypatch, current=6, initialized=False)
    assert engine.resolve_device() == 1
    # a pinned default beats both
    engine.set_default_device(2)
    try:
        assert engine.resolve_device() == 2
        assert engine.resolve_device(7) == 7
    finally:
        engine.set_default_device(None)


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_gpus_flag_starts_the_ranks_as_a_child(monkeypatch):
    """``python bench.py --gpus 4`` (no torchrun environment) must start 4 ranks itself -- as a child
    process, before this process imports torch or touches a GPU -- and exit with the child's code."""
    bench = _load_bench()
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    had_torch = "torch" in sys.modules
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 7
    assert ("torch" in sys.modules) == had_torch             # the parent did not import torch on the way
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert str(ROOT / "bench.py") in cmd
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_under_torchrun_does_not_spawn_again(monkeypatch):
    bench = _load_bench()
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: pytest.fail("a rank must not start ranks"))
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])

    class Stop(Exception):
        pass

    # the first thing a rank does after the decision is importing torch: stop there
    real_import = __import__

    def guarded(name, *a, **k):
        if name == "torch":
            raise Stop
        return real_import(name, *a, **k)

    monkeypatch.setattr("builtins.__import__", guarded)
    with pytest.raises(Stop):
        bench.main()
