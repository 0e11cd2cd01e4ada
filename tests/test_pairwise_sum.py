"""numpy's float64 add.reduce order (pairwise blocks of <= 128, eight accumulators, 8192-element
buffers) as the engine restates it -- the order behind the mean of the k-NN distances
(utilities.py:1434) and of the ray exits (utilities.py:1650).  Oracle: numpy itself."""
import ctypes

import numpy as np
import pytest

# lengths around every structural boundary of the recursion: the 8-element unroll, the 128-element leaf,
# the halving (n/2 rounded down to a multiple of 8), the 8192-element buffer of the outer iterator
LENGTHS = sorted(set(
    list(range(0, 20)) + [63, 64, 65, 120, 127, 128, 129, 135, 136, 137, 255, 256, 257, 263, 264, 519, 520, 1031,
                          2055, 4095, 4096, 4103, 7980, 8191, 8192, 8193, 8199, 8200, 9470, 12345, 16383, 16384, 16385,
                          20480]))


def _values(n, seed):
    rng = np.random.default_rng(seed)
    # magnitudes spread over eight decades so that a wrong association shows in the last bits
    return rng.normal(size=n) * 10.0 ** rng.integers(-4, 4, size=n)


@pytest.fixture(scope="module")
def probe(hostsim):
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    for f in (L.hs_pairwise_sum, L.hs_pairwise_sum_serial, L.hs_pairwise_sum_lean):
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_void_p, ctypes.c_long]
    return L


def test_host_team_sum_is_numpy_sum(probe):
    for n in LENGTHS:
        a = _values(n, n)
        want = float(np.add.reduce(a)) if n else 0.0
        assert probe.hs_pairwise_sum(a.ctypes.data, n) == want, n
        assert probe.hs_pairwise_sum_serial(a.ctypes.data, n) == want, n
        if n:
            assert probe.hs_pairwise_sum_lean(a.ctypes.data, n) == want, n      # (the scalar-state walk of the re-assembly)


@pytest.mark.gpu
@pytest.mark.parametrize("one_wave", [False, True], ids=["four waves", "one wave"])
@pytest.mark.parametrize("global_scratch", [False, True], ids=["scratch in LDS", "scratch in global memory"])
def test_gpu_team_sum_is_numpy_sum(one_wave, global_scratch):
    from pywindow_amd import _lib
    ctx = _lib.Context(0)
    try:
        for n in LENGTHS:
            a = _values(n, 1000 + n)
            want = float(np.add.reduce(a)) if n else 0.0
            got = ctx.pairwise_sum(a, one_wave=one_wave, global_scratch=global_scratch)
            assert got == want, (n, got, want)
    finally:
        ctx.close()
