"""Sampling sphere of find_windows over its whole range (rows a-9a): number of sampling vectors,
sphere radius and the DBSCAN radius ``eps`` (mean of all 10-nearest-neighbour distances, utilities.py:
1399-1434) for sphere radii from 3.6 to 3000 Angstrom and three ``adjust`` factors, i.e. 16 ... 2048
sampling vectors -- the cage fixtures only know ~800.  The k-NN search of the kernel is written for the
golden spiral (index windows that prove themselves, a full scan where they cannot): this is where its
corners are.  Molecules: two carbon atoms at the distance that gives the radius.  Oracle: the CPU
restatement (scipy's KDTree on numpy's points), run live."""
import ctypes
import sys

import numpy as np
import pytest

from _util import LIVE_TOL_WINDOW, ROOT

sys.path.insert(0, str(ROOT))
from oracle import pw_oracle as O  # noqa: E402  (test infrastructure: the checker)

C_VDW, C_MASS = 1.70, 12.011
RADII = np.unique(np.concatenate([np.geomspace(3.6, 3000.0, 36), [5.0, 10.5, 64.0, 1000.0]]))   # (below 3.4 the pore would be negative)
ADJUST = (1.0, 0.31, 0.085)


def _batch(radii):
    xyz = np.zeros((len(radii), 2, 3))
    xyz[:, 1, 0] = 2.0 * radii - 2.0 * C_VDW          # max_dim = distance + both radii = 2 * radius
    off = np.arange(len(radii) + 1, dtype=np.int64) * 2
    return off, np.ascontiguousarray(xyz.reshape(-1, 3)), np.full(2 * len(radii), C_VDW), np.full(2 * len(radii), C_MASS)


def _check(recs, radii, adjust, where):
    seen = set()
    for u, rec in enumerate(recs):
        r = float(rec["sphere_r"])
        assert r == (float(2.0 * radii[u] - 2.0 * C_VDW) + (C_VDW + C_VDW)) / 2.0, (where, u)
        count = O.n_sampling_points(r, adjust)
        assert int(rec["n_points"]) == count, (where, u, r)
        if count < 10:                                      # the reference raises there (KDTree.query(k=10)): flagged
            assert int(rec["status"]) & 64, (where, u, count)
            continue
        assert not int(rec["status"]) & 4, (where, u, count)     # (no capacity is exceeded: the workspace follows adjust)
        want = O.knn_eps(O.sphere_points(r, count))
        got = float(rec["eps"])
        assert abs(got - want) <= LIVE_TOL_WINDOW * abs(want), (where, u, r, count, got, want)
        seen.add(count)
    return seen


def test_host_team_sampling_sphere(hostsim):
    from pywindow_amd import _lib
    # one third of the radii on the host (the one-thread team is slow for 2000 vectors); the GPU takes all
    radii = RADII[::3]
    off, xyz, vdw, mass = _batch(radii)
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    out = np.zeros(len(radii), dtype=_lib.UNIT_OUT_DTYPE)
    vp = ctypes.c_void_p
    rc = L.hs_analysis_batch(ctypes.c_long(len(radii)), off.ctypes.data_as(vp), xyz.ctypes.data_as(vp),
                             vdw.ctypes.data_as(vp), mass.ctypes.data_as(vp), ctypes.c_uint(15),
                             out.ctypes.data_as(vp), None)
    assert rc == 0
    assert len(_check(out, radii, 1.0, "host")) >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("adjust", ADJUST)
def test_gpu_sampling_sphere(hip_ctx, adjust):
    from pywindow_amd import _lib
    off, xyz, vdw, mass = _batch(RADII)
    try:
        recs = hip_ctx.analyse(_lib.Batch(off, xyz, vdw, mass), params=_lib.Params(adjust_windows=adjust))
    finally:
        hip_ctx.set_params(None)
    seen = _check(recs, RADII, adjust, f"gpu adjust {adjust}")
    assert len(seen) >= 20
    if adjust == 1.0:
        assert max(seen) > 1900
    if adjust == min(ADJUST):
        assert min(seen) < 60
