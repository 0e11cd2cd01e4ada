"""DBSCAN of the window search on its own (row a-9d): ``sklearn.cluster.DBSCAN(eps, min_samples=5)`` as
find_windows calls it (utilities.py:1478-1487).  Oracle: scikit-learn itself, on point sets built to
exercise what the cage fixtures do not -- border points between clusters (their label depends on the
ORDER in which sklearn grows the clusters), noise, more points than threads, a single cluster, none.  Every set also in
order of falling z, as the window search hands its survivors over: the adjacency rows then skip blocks of candidates
that are out of reach in z (round 6), which an unordered cloud never exercises."""
import ctypes

import numpy as np
import pytest

sklearn_cluster = pytest.importorskip("sklearn.cluster")


def _cloud(n, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "blobs":            # tight blobs + uniform background: cores, borders and noise
        k = max(1, n // 40)
        centres = rng.uniform(-10, 10, size=(k, 3))
        pts = centres[rng.integers(0, k, size=n)] + rng.normal(0.0, 0.6, size=(n, 3))
        m = rng.random(n) < 0.25
        pts[m] = rng.uniform(-12, 12, size=(int(m.sum()), 3))
        eps = 0.9
    elif kind == "chain":          # one long thin cluster: many rounds of label propagation
        t = np.sort(rng.uniform(0, 1, size=n))
        pts = np.stack([40 * t, np.sin(12 * t), np.cos(7 * t)], 1) + rng.normal(0.0, 0.05, size=(n, 3))
        eps = 40.0 / n * 4.0
    elif kind == "bridges":        # dense clumps joined by sparse bridges of non-core points
        k = 6
        centres = np.stack([np.arange(k) * 3.0, np.zeros(k), np.zeros(k)], 1)
        pts = centres[rng.integers(0, k, size=n)] + rng.normal(0.0, 0.35, size=(n, 3))
        b = rng.random(n) < 0.15
        pts[b] = np.stack([rng.uniform(0, 3.0 * (k - 1), size=int(b.sum())), rng.normal(0, 0.1, int(b.sum())),
                           rng.normal(0, 0.1, int(b.sum()))], 1)
        eps = 0.55
    elif kind == "sparse":         # nothing is a core point
        pts = rng.uniform(-50, 50, size=(n, 3))
        eps = 0.5
    else:                          # "lattice": many exactly equal distances (ties at the radius)
        side = int(round(n ** (1 / 3))) + 1
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(float)
        pts = g[rng.permutation(len(g))[:n]]
        eps = 1.0
    return np.ascontiguousarray(pts[:n]), float(eps)


CASES = [(n, kind) for kind in ("blobs", "chain", "bridges", "sparse", "lattice")
         for n in (1, 4, 5, 6, 63, 64, 65, 200, 255, 256, 257, 300, 511, 513, 900, 2048)]
# beyond the 2048 points the workspaces were limited to until round 3 (find_windows with adjust > 2.5 gets there)
CASES += [(2049, "blobs"), (3001, "bridges"), (5000, "lattice"), (8192, "blobs")]


def _by_falling_z(pts):
    return np.ascontiguousarray(pts[np.argsort(-pts[:, 2], kind="stable")])


def _sklearn_labels(pts, eps):
    return sklearn_cluster.DBSCAN(eps=eps, min_samples=5).fit(pts).labels_.astype(np.int32)


def test_host_team_dbscan_is_sklearn(hostsim):
    L = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    L.hs_dbscan.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_double, ctypes.c_void_p]
    kinds = set()
    for n, kind in CASES:
        cloud, eps = _cloud(n, kind, 7 * n + len(kind))
        for pts in (cloud, _by_falling_z(cloud)):
            want = _sklearn_labels(pts, eps)
            got = np.zeros(n, dtype=np.int32)
            k = L.hs_dbscan(pts.ctypes.data, n, eps, got.ctypes.data)
            assert np.array_equal(got, want), (n, kind)
            assert k == (want.max() + 1 if (want >= 0).any() else 0), (n, kind)
            if (want >= 0).any() and (want < 0).any():
                kinds.add(kind)
    assert {"blobs", "bridges"} <= kinds          # (the point sets do mix clusters and noise)


@pytest.mark.gpu
@pytest.mark.parametrize("one_wave", [False, True], ids=["four waves", "one wave"])
@pytest.mark.parametrize("global_memory", [False, True], ids=["arrays in LDS", "arrays in global memory"])
def test_gpu_team_dbscan_is_sklearn(one_wave, global_memory):
    from pywindow_amd import _lib
    ctx = _lib.Context(0)
    try:
        for n, kind in CASES:
            cloud, eps = _cloud(n, kind, 7 * n + len(kind))
            for ordered, pts in ((False, cloud), (True, _by_falling_z(cloud))):
                want = _sklearn_labels(pts, eps)
                for rep in range(2):           # (the propagation is racy by design: the fixed point must not be)
                    got, k = ctx.dbscan(pts, eps, one_wave=one_wave, global_memory=global_memory)
                    assert np.array_equal(got, want), (n, kind, ordered, rep)
                    assert k == (want.max() + 1 if (want >= 0).any() else 0), (n, kind, ordered)
    finally:
        ctx.close()
