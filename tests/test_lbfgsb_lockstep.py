"""pw::Lbfgsb<N> against SciPy's own L-BFGS-B engine, call by call: x, the task
code and the whole workspace (ws, wy, sy, ss, wt, wn, z, r, d) must agree bit for
bit after EVERY setulb call (tests/tools/lockstep_lbfgsb.py)."""
import sys

import numpy as np
import pytest

pytest.importorskip("scipy.optimize")
from _util import ROOT  # noqa: E402

sys.path.insert(0, str(ROOT / "tests" / "tools"))


@pytest.fixture(scope="module")
def LS(hostsim):
    import lockstep_lbfgsb

    return lockstep_lbfgsb


@pytest.mark.parametrize("tag,units", [("md20", (1, 13, 14, 16)), ("static", (0, 5, 10)), ("synth64", (0, 7))])
def test_pore_centre_runs_are_lockstep_identical(LS, tag, units):
    from oracle import pw_oracle as O

    g = np.load(ROOT / f"tests/golden/{tag}.npz")
    for u in units:
        cage = LS.cage_from_fixture(g, u)
        com = O.centre_of_mass(cage)
        r = O.pore_diameter(cage, com)[0] / 2
        res = LS.lockstep(lambda c: -(cage.gap(c)[0] * 2), com, com - r, com + r, [2, 2, 2])
        assert res["first_bad"] is None, (tag, u, res["first_bad"])
        assert np.array_equal(res["mx"], g["st_opt_x"][u])
        assert res["nit"] == g["st_opt_nit"][u]


def test_window_neck_runs_are_lockstep_identical(LS):
    n = 0
    for tag in ("static", "md20"):
        for u, w, cage, zlb, zx in LS.z_problems(tag):
            res = LS.lockstep(lambda z: cage.gap(np.array([0.0, 0.0, z[0]]))[0] * 2, np.array([0.0]),
                              np.array([zlb]), np.array([np.inf]), [1])
            assert res["first_bad"] is None and res["mx"][0] == zx, (tag, u, w)
            n += 1
            if n >= 24:
                return
