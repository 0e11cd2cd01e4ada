"""Lbfgsb::minimize (the direct form that every product path runs since round 5) against the reverse-communication
loop around Lbfgsb::step (the form the lockstep tests hold against SciPy's own setulb call by call): the two drivers on
the SAME objectives must end with the same bits -- x, f, iteration and evaluation counts, task and message codes, and
every array of the optimiser's state.  (Round 5's advisor: the lockstep guarantee covered step(), the product ran
minimize(), and nothing compared the two.)  Objectives: the pore-centre problem (N = 3) and the window-neck problem
(N = 1) of golden-fixture molecules, boxes small enough that the iterate runs along its bounds, an objective whose
gradient lies (the line search fails: ABNORMAL / restarts), a flat one (first projected gradient below pgtol), and
iteration / evaluation limits that stop a run half way."""
import ctypes
import sys

import numpy as np
import pytest

pytest.importorskip("scipy.optimize")
from _util import ROOT  # noqa: E402

sys.path.insert(0, str(ROOT / "tests" / "tools"))
M = 10
CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                      ctypes.c_void_p)


@pytest.fixture(scope="module")
def probe(hostsim):
    L = ctypes.CDLL(str(ROOT / "tests" / "hostsim" / "liblbprobe.so"))
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    for n in (1, 2, 3):
        getattr(L, f"hs_lb{n}_drive").argtypes = [ctypes.c_int, dp, dp, dp, ip, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                  CB, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ip, dp, ip, dp]
    return L


def drive(L, mode, fun_and_grad, x0, lb, ub, nbd, per_call=1, maxiter=15000, maxfun=15000):
    n = len(x0)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    calls = []

    def cb(xp, fp, gp, _user):
        x = np.array([xp[i] for i in range(n)])
        f, g = fun_and_grad(x)
        calls.append(x)
        fp[0] = f
        for i in range(n):
            gp[i] = g[i]

    low = np.where(np.isinf(lb), 0.0, lb).astype(float)
    up = np.where(np.isinf(ub), 0.0, ub).astype(float)
    nb = np.array(nbd, np.int32)
    tot = 2 * M * n + 3 * M * M + 8 * M * M + 5 * n + 8 * M
    x_out, f_out, oi = np.zeros(n), np.zeros(1), np.zeros(4, np.int32)
    wa, ints, dbl = np.zeros(tot), np.zeros(16, np.int32), np.zeros(16)
    P = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    getattr(L, f"hs_lb{n}_drive")(mode, P(np.array(x0, float)), P(low), P(up), nb.ctypes.data_as(ip), 1e7, 1e-5, 20, CB(cb), None,
                                  per_call, maxiter, maxfun, P(x_out), P(f_out), oi.ctypes.data_as(ip), P(wa), ints.ctypes.data_as(ip),
                                  P(dbl))
    return {"x": x_out, "f": f_out[0], "nit": int(oi[0]), "nfev": int(oi[1]), "task": int(oi[2]), "msg": int(oi[3]), "wa": wa,
            "ints": ints.copy(), "dbl": dbl.copy(), "calls": calls}


def same(a, b, what):
    assert a["nit"] == b["nit"] and a["nfev"] == b["nfev"], (what, a["nit"], b["nit"], a["nfev"], b["nfev"])
    assert (a["task"], a["msg"]) == (b["task"], b["msg"]), (what, a["task"], a["msg"], b["task"], b["msg"])
    assert a["x"].tobytes() == b["x"].tobytes(), (what, a["x"], b["x"])
    assert np.float64(a["f"]).tobytes() == np.float64(b["f"]).tobytes(), what
    assert len(a["calls"]) == len(b["calls"]) and all(p.tobytes() == q.tobytes() for p, q in zip(a["calls"], b["calls"])), what
    # the whole state: correction pairs, the middle matrices, z / r / d, the counters and the scalars of the search
    # (xp, t and the work array wa are scratch that the direct form keeps in registers: not compared)
    n = len(a["x"])
    cut = 2 * M * n + 3 * M * M + 8 * M * M + 3 * n
    assert a["wa"][:cut].tobytes() == b["wa"][:cut].tobytes(), (what, np.nonzero(a["wa"][:cut] != b["wa"][:cut])[0][:8])
    assert np.array_equal(a["ints"][:13], b["ints"][:13]), (what, a["ints"][:13], b["ints"][:13])
    assert a["dbl"][:9].tobytes() == b["dbl"][:9].tobytes(), (what, a["dbl"][:9], b["dbl"][:9])


def both(L, fun_and_grad, x0, lb, ub, nbd, what, **kw):
    a = drive(L, 0, fun_and_grad, x0, lb, ub, nbd, **kw)
    b = drive(L, 1, fun_and_grad, x0, lb, ub, nbd, **kw)
    same(a, b, what)
    return a


def fd(fun, lb, ub):
    import lockstep_lbfgsb as LS

    def fg(x):
        f = fun(x)
        return f, LS.fd_grad(fun, x, f, lb, ub)

    return fg


@pytest.mark.parametrize("tag,units", [("md20", (1, 13, 14)), ("static", (0, 5, 10)), ("synth64", (0, 7, 21))])
def test_pore_centre_problems(probe, tag, units):
    import lockstep_lbfgsb as LS
    from oracle import pw_oracle as O

    g = np.load(ROOT / f"tests/golden/{tag}.npz")
    for u in units:
        cage = LS.cage_from_fixture(g, u)
        com = O.centre_of_mass(cage)
        r = O.pore_diameter(cage, com)[0] / 2
        fun = lambda c: -(cage.gap(c)[0] * 2)  # noqa: E731
        res = both(probe, fd(fun, com - r, com + r), com, com - r, com + r, [2, 2, 2], (tag, u), per_call=4)
        assert np.array_equal(res["x"], g["st_opt_x"][u]) and res["nit"] == g["st_opt_nit"][u]      # (... and the reference's)
        # a box a tenth the size: the iterate ends on its bounds, variables enter and leave the free set
        small = r / 10
        both(probe, fd(fun, com - small, com + small), com, com - small, com + small, [2, 2, 2], (tag, u, "small box"), per_call=4)
        # limits that stop the run at a new iterate (SciPy tests them there only)
        both(probe, fd(fun, com - r, com + r), com, com - r, com + r, [2, 2, 2], (tag, u, "maxiter"), per_call=4, maxiter=3)
        both(probe, fd(fun, com - r, com + r), com, com - r, com + r, [2, 2, 2], (tag, u, "maxfun"), per_call=4, maxfun=40)


def test_window_neck_problems(probe):
    import lockstep_lbfgsb as LS

    n = 0
    for tag in ("static", "md20"):
        for u, w, cage, zlb, zx in LS.z_problems(tag):
            fun = lambda z: cage.gap(np.array([0.0, 0.0, z[0]]))[0] * 2  # noqa: E731
            lb, ub = np.array([zlb]), np.array([np.inf])
            res = both(probe, fd(fun, lb, ub), np.array([0.0]), lb, ub, [1], (tag, u, w), per_call=2)
            assert res["x"][0] == zx
            n += 1
            if n >= 16:
                return


def test_hard_cases(probe):
    rng = np.random.default_rng(5)
    # a gradient that lies (points uphill): the first line search cannot find a step -- ABNORMAL with no correction
    # pair, a restart (refresh) with some
    quad = lambda x: float(np.sum((x - 0.3) ** 2))  # noqa: E731
    lying = lambda x: (quad(x), -2.0 * (x - 0.3))  # noqa: E731
    lb, ub = np.full(3, -2.0), np.full(3, 2.0)
    both(probe, lying, np.array([1.0, -1.0, 0.5]), lb, ub, [2, 2, 2], "lying gradient")
    k = [0]

    def lying_later(x):
        k[0] += 1
        g = 2.0 * (x - 0.3) * np.array([1.0, 30.0, 0.2])
        f = float(np.sum((x - 0.3) ** 2 * np.array([1.0, 30.0, 0.2])))
        return f, (g if k[0] <= 6 else -g)

    for mode_first in (0, 1):          # (the call counter belongs to a run: fresh for each driver)
        k[0] = 0
        a = drive(probe, mode_first, lying_later, np.array([1.5, -1.0, 0.5]), lb, ub, [2, 2, 2])
        k[0] = 0
        b = drive(probe, 1 - mode_first, lying_later, np.array([1.5, -1.0, 0.5]), lb, ub, [2, 2, 2])
        same(a, b, "gradient that starts lying after six evaluations")
    # flat: converged at the start point
    both(probe, lambda x: (1.0, np.zeros(3)), np.zeros(3), lb, ub, [2, 2, 2], "flat")
    # ill-conditioned quadratics in boxes that cut the minimum off, one / two / three variables, every bound kind
    for trial in range(40):
        n = int(rng.integers(1, 4))
        scale = 10.0 ** rng.uniform(-2, 3, size=n)
        centre = rng.normal(size=n)
        A = rng.normal(size=(n, n))
        H = A @ A.T + np.diag(scale)
        fg = lambda x, H=H, centre=centre: (float(0.5 * (x - centre) @ H @ (x - centre)), H @ (x - centre))  # noqa: E731
        lo_, hi_ = centre - rng.uniform(-0.5, 2.0, size=n), centre + rng.uniform(-0.5, 2.0, size=n)
        lo_, hi_ = np.minimum(lo_, hi_), np.maximum(lo_, hi_)
        nbd = rng.integers(0, 4, size=n)
        lbv = np.where((nbd == 1) | (nbd == 2), lo_, -np.inf)
        ubv = np.where((nbd == 2) | (nbd == 3), hi_, np.inf)
        x0 = np.clip(centre + rng.normal(size=n) * 3, np.where(np.isinf(lbv), -1e30, lbv), np.where(np.isinf(ubv), 1e30, ubv))
        both(probe, fg, x0, lbv, ubv, list(nbd), ("quadratic", trial))
