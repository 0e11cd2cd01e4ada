"""pw_math.hpp: sin/cos/arccos correctly rounded (vs mpmath when present), and within
one ulp of numpy's values on the arguments the path uses."""
import ctypes

import numpy as np
import pytest

DP = ctypes.POINTER(ctypes.c_double)


def P(a):
    return a.ctypes.data_as(DP)


@pytest.fixture(scope="module")
def L(hostsim):
    return ctypes.CDLL(str(hostsim / "libmathprobe.so"))


def test_sincos_on_golden_spiral_angles(L):
    th = np.pi * (3 - np.sqrt(5)) * np.arange(2500)
    s = np.empty_like(th)
    c = np.empty_like(th)
    L.hs_sincos(len(th), P(th), P(s), P(c))
    assert np.max(np.abs(s - np.sin(th)) / np.spacing(np.abs(np.sin(th)))) <= 1.0
    assert np.max(np.abs(c - np.cos(th)) / np.spacing(np.abs(np.cos(th)))) <= 1.0
    assert (s != np.sin(th)).mean() < 0.01 and (c != np.cos(th)).mean() < 0.01
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    ref_s = np.array([float(mp.sin(mp.mpf(float(t)))) for t in th[:600]])
    ref_c = np.array([float(mp.cos(mp.mpf(float(t)))) for t in th[:600]])
    assert np.array_equal(s[:600], ref_s) and np.array_equal(c[:600], ref_c)


def test_acos_and_log10(L):
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(0, 1, 5000), 1 - 10.0 ** rng.uniform(-16, -1, 500), [0.0, 1.0]])
    y = np.empty_like(a)
    L.hs_acos01(len(a), P(a), P(y))
    ref = np.arccos(a)
    assert np.max(np.abs(y - ref) / np.maximum(np.spacing(ref), 1e-300)) <= 1.0
    area = rng.uniform(100, 60000, 20000)
    lg = np.empty_like(area)
    L.hs_log10(len(area), P(area), P(lg))
    assert np.array_equal(np.floor(lg * 250), np.floor(np.log10(area) * 250))
