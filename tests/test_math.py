"""pw_math.hpp: numpy's sin / cos / arccos / scalar ** reproduced bit for bit (glibc 2.35 s_sin.c
and e_pow.c, Intel SVML __svml_acos8_ha with the VRSQRT14PD table)."""
import ctypes

import numpy as np
import pytest

DP = ctypes.POINTER(ctypes.c_double)


def P(a):
    return a.ctypes.data_as(DP)


@pytest.fixture(scope="module")
def L(hostsim):
    return ctypes.CDLL(str(hostsim / "libmathprobe.so"))


def test_sincos_equal_the_c_library(L):
    """numpy.sin / numpy.cos (glibc s_sin.c) reproduced bit for bit: the golden-spiral angles of
    the sampling spheres, every branch of the range reduction, arguments near multiples of pi/2."""
    import math
    import platform

    rng = np.random.default_rng(0)
    sets = [np.pi * (3 - np.sqrt(5)) * np.arange(5000), rng.uniform(-0.126, 0.126, 200000),
            rng.uniform(-0.8555, 0.8555, 200000), rng.uniform(0.85, 2.43, 200000) * rng.choice([-1, 1], 200000),
            rng.uniform(2.4, 7, 200000) * rng.choice([-1, 1], 200000), rng.uniform(7, 3000, 200000),
            rng.uniform(3000, 1.05e8, 200000), rng.uniform(-1e-7, 1e-7, 20000),
            (np.arange(1, 2001) * np.pi / 2)[:, None].repeat(20, 1).ravel() + rng.normal(0, 1e-9, 40000)]
    exact = platform.libc_ver() == ("glibc", "2.35")
    for th in sets:
        s = np.empty_like(th)
        c = np.empty_like(th)
        L.hs_sincos(len(th), P(th), P(s), P(c))
        rs, rc = np.sin(th), np.cos(th)
        assert np.max(np.abs(s - rs) / np.spacing(np.abs(rs))) <= 1.0
        assert np.max(np.abs(c - rc) / np.spacing(np.abs(rc))) <= 1.0
        if exact:   # other C libraries round differently in the last bit
            assert np.array_equal(s, rs) and np.array_equal(c, rc)
            assert all(float(s[k]) == math.sin(float(th[k])) for k in range(0, len(th), 997))


def test_sine_is_odd_and_cosine_even_bit_for_bit(L):
    """wave_window takes the sine and cosine of -a from those of a (the back-rotation of a window's centre): true of the
    C library -- numpy's scalar sin / cos -- and of the restatement, bit for bit, over the range of the rotation angles
    (|a| <= 5 pi / 2) and beyond."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 8, 300000), rng.uniform(8, 3000, 100000), rng.uniform(0, 1e-6, 1000),
                        np.arange(1, 41) * np.pi / 4])
    assert np.array_equal(np.sin(-x), -np.sin(x)) and np.array_equal(np.cos(-x), np.cos(x))
    for th in (x, -x):
        s = np.empty_like(th)
        c = np.empty_like(th)
        L.hs_sincos(len(th), P(th), P(s), P(c))
        if th is x:
            s_pos, c_pos = s, c
    assert np.array_equal(s, -s_pos) and np.array_equal(c, c_pos)
    assert np.array_equal(np.signbit(s), ~np.signbit(s_pos) | (s_pos != s_pos))


def test_scalar_power_equals_the_c_library(L):
    """float64 scalar ** in numpy is glibc's pow(): x ** 2 is NOT always x * x."""
    import math
    import platform

    L.hs_pow.argtypes = [ctypes.c_int, DP, ctypes.c_double, DP]
    rng = np.random.default_rng(1)
    exact = platform.libc_ver() == ("glibc", "2.35")
    for y in (2.0, 3.0, 0.5):
        x = np.concatenate([rng.uniform(1e-3, 30, 100000), rng.uniform(0.5, 1e4, 50000)])
        out = np.empty_like(x)
        L.hs_pow(len(x), P(x), y, P(out))
        ref = np.array([math.pow(float(v), y) for v in x])
        assert np.max(np.abs(out - ref) / np.spacing(ref)) <= 1.0
        if exact:
            assert np.array_equal(out, ref)
            assert all(np.float64(v) ** y == r for v, r in zip(x[:2000], ref[:2000]))
    x = rng.uniform(1e-3, 30, 200000)
    sq = np.array([math.pow(float(v), 2.0) for v in x])
    assert 0 < (sq != x * x).sum() < 1000        # the reason this function exists


def test_acos_and_log10(L):
    rng = np.random.default_rng(0)
    # numpy's arccos (SVML __svml_acos8_ha + the VRSQRT14PD table) is reproduced bit for bit
    a = np.concatenate([rng.uniform(0, 1, 400000), rng.uniform(-1, 1, 100000), 1 - 10.0 ** rng.uniform(-16, -1, 20000),
                        0.5 + rng.normal(0, 1e-3, 20000), [0.0, 1.0, -1.0, 0.5, -0.5, 0.25, 1e-300, 0.7071067811865476]])
    y = np.empty_like(a)
    L.hs_acos(len(a), P(a), P(y))
    ref = np.arccos(a)
    # on a host whose numpy has no AVX-512 SVML path (plain libm arccos) only one ulp is guaranteed
    assert np.max(np.abs(y - ref) / np.maximum(np.spacing(ref), 1e-300)) <= 1.0
    from numpy._core._multiarray_umath import __cpu_features__ as feats

    if feats.get("AVX512_SKX"):
        assert np.array_equal(y, ref)
    area = rng.uniform(100, 60000, 20000)
    lg = np.empty_like(area)
    L.hs_log10(len(area), P(area), P(lg))
    assert np.array_equal(np.floor(lg * 250), np.floor(np.log10(area) * 250))
