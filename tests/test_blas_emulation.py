"""pywindow_amd/csrc/pw_blas.hpp + pw_ext64.hpp against the OpenBLAS SciPy links,
bit for bit, on the tiny shapes L-BFGS-B uses."""
import ctypes

import numpy as np
import pytest

scipy_linalg = pytest.importorskip("scipy.linalg")
from scipy.linalg import blas, lapack  # noqa: E402

DP = ctypes.POINTER(ctypes.c_double)


def P(a):
    return a.ctypes.data_as(DP)


@pytest.fixture(scope="module")
def L(hostsim):
    lib = ctypes.CDLL(str(hostsim / "libblasprobe.so"))
    lib.hs_ddot.restype = ctypes.c_double
    lib.hs_dnrm2.restype = ctypes.c_double
    return lib


def test_ddot_daxpy(L):
    rng = np.random.default_rng(0)
    for n in range(1, 32):
        for _ in range(200):
            x = rng.normal(size=n) * 10 ** rng.uniform(-3, 3)
            y = rng.normal(size=n)
            assert blas.ddot(x, y) == L.hs_ddot(n, P(x), P(y))
            a = rng.normal()
            y2 = y.copy()
            L.hs_daxpy(n, ctypes.c_double(a), P(x), P(y2))
            assert np.array_equal(blas.daxpy(x, y.copy(), a=a), y2)


def test_dnrm2_x87(L):
    rng = np.random.default_rng(1)
    for _ in range(20000):
        n = int(rng.integers(1, 8))
        x = rng.normal(size=n) * 10 ** rng.uniform(-8, 3, size=n)
        assert blas.dnrm2(x) == L.hs_dnrm2(n, P(x))


def test_dnrm2_short_route_equals_the_emulation(L):
    """b_dnrm2 takes a double-double route when its result is provably the x87 one and the integer
    emulation otherwise: both on random vectors and on vectors whose norm sits within 1e-9 ulp of a
    midpoint between two doubles (where the three roundings of the x87 code decide the result)."""
    L.hs_dnrm2_exact.restype = ctypes.c_double
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 3, 5, 8):
        m = 100000
        x = np.ascontiguousarray(rng.normal(size=(m, n)) * 10 ** rng.uniform(-8, 3, size=(m, n)))
        fast, exact = np.zeros(m), np.zeros(m)
        L.hs_dnrm2_many(ctypes.c_long(m), n, P(x), P(fast), P(exact))
        assert np.array_equal(fast, exact), n
    m = 100000
    y = rng.uniform(1, 2, size=m)
    x1 = np.sqrt(y * np.spacing(y) * (1 + rng.uniform(-1e-9, 1e-9, size=m)))
    x = np.ascontiguousarray(np.stack([y, x1, np.zeros(m)], axis=1))
    fast, exact = np.zeros(m), np.zeros(m)
    L.hs_dnrm2_many(ctypes.c_long(m), 3, P(x), P(fast), P(exact))
    assert np.array_equal(fast, exact)
    for i in range(2000):
        assert blas.dnrm2(x[i]) == fast[i]
    assert L.hs_dnrm2(3, P(np.zeros(3))) == 0.0


def test_dpotrf(L):
    rng = np.random.default_rng(2)
    for n in range(1, 11):
        for lda in (10, 20):
            for _ in range(60):
                A = rng.normal(size=(n + 3, n))
                S = A.T @ A + np.eye(n) * 0.1
                F = np.zeros((lda, lda), order="F")
                F[:n, :n] = S
                ref, info = lapack.dpotrf(np.asfortranarray(S), lower=0, clean=0)
                assert L.hs_dpotrf_u(n, P(F), lda) == info
                assert np.array_equal(np.triu(ref), np.triu(F[:n, :n]))


def test_dtrtrs(L):
    rng = np.random.default_rng(3)
    for n in range(1, 21):
        for _ in range(60):
            U = np.asfortranarray(np.triu(rng.normal(size=(n, n))) + np.eye(n) * 3)
            b = rng.normal(size=n)
            for tr in (0, 1):
                ref, _ = lapack.dtrtrs(U, b, lower=0, trans=tr)
                x = b.copy()
                L.hs_dtrtrs_u(tr, n, 1, P(U), n, P(x), n)
                assert np.array_equal(ref, x)
            if 2 <= n <= 10:
                B = np.asfortranarray(rng.normal(size=(n, n)))
                ref, _ = lapack.dtrtrs(U, B, lower=0, trans=1)
                X = B.copy(order="F")
                L.hs_dtrtrs_u(1, n, n, P(U), n, P(X), n)
                assert np.array_equal(ref, X)
