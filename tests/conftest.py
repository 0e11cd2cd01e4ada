import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device); run with -m gpu")


def pytest_report_header(config):
    """Which branch of the live-oracle comparisons this run takes (tests/_util.py): bit-exact on the reference
    platform (glibc 2.35 libm, AVX-512 numpy -- what the golden fixtures were produced on), north_star's 1e-6
    elsewhere, where the oracle's own sin / cos / pow / arccos round differently.  Golden-fixture comparisons are
    tolerance 0 on every platform."""
    import platform

    sys.path.insert(0, str(ROOT / "tests"))
    import _util

    libc = " ".join(platform.libc_ver())
    if _util.LIVE_TOL_WINDOW == 0.0:
        return [f"pywindow_amd parity: reference platform ({libc}, AVX-512 numpy) -- live-oracle comparisons at tolerance 0"]
    return [f"pywindow_amd parity: NOT the reference platform ({libc}) -- live-oracle comparisons relaxed to "
            f"{_util.LIVE_TOL_WINDOW:g} relative (golden fixtures stay at 0)"]


@pytest.fixture(scope="session")
def hostsim():
    """CPU-only harnesses that compile the kernel headers with a one-thread team."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("hostsim_build", ROOT / "tests" / "hostsim" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    return ROOT / "tests" / "hostsim"


@pytest.fixture(scope="session")
def hip_ctx():
    """THE context of device 0 in this process -- the one `engine.context()` hands to the façade as well: one
    context per device keeps the hardware queues for its ten streams (a second live context would run single
    launches, see pw_context_create's stream probe), so the parity tests exercise the overlapped pipeline."""
    from pywindow_amd import engine

    return engine.context(0)
