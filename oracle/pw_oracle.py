"""CPU oracle for the pywindow ``full_analysis()`` hot path.  TEST INFRASTRUCTURE.

This module is the checker, never the thing shipped or measured as the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  ``pywindow_amd`` never does.

It restates the reference's algorithm (numpy + scipy.optimize + scikit-learn,
the reference's own third-party stack) with the distance primitive replaced by
the bit-exact scalar recipe in ``oracle/pw_prim.c`` so that results do not depend
on which OpenBLAS kernels the host CPU selects.  Each function cites the
reference lines it follows (paths relative to
``/root/reference/src/pywindow/_internal/``).

PARITY PIN: checked against golden vectors produced by the reference itself in
the development container -- tests/golden/*.npz made by
tests/golden/make_golden.py (11 literal inputs of the reference's own tests,
20 real MD frames, 64 synthetic frames, 8 cages of the periodic cell), see
tests/test_oracle.py.
"""

from __future__ import annotations

import ctypes
import pathlib
import subprocess

import numpy as np
from scipy.optimize import brute, fmin, minimize
from sklearn.cluster import DBSCAN
from sklearn.neighbors import KDTree

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None
W_MAX = 16


def build(force: bool = False) -> pathlib.Path:
    """Compile oracle/pw_prim.c -> oracle/libpworacle.so (gcc, no contraction)."""
    so = _HERE / "libpworacle.so"
    src = _HERE / "pw_prim.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-mfma", "-fPIC", "-shared",
             "-o", str(so), str(src), "-lm"]
        )
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(str(build()))
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int64)
        L.pwo_row_sqnorms.argtypes = [ctypes.c_int64, dp, dp]
        L.pwo_min_gap.argtypes = [ctypes.c_int64, dp, dp, dp, dp, ip]
        L.pwo_min_gap.restype = ctypes.c_double
        L.pwo_max_dim.argtypes = [ctypes.c_int64, dp, dp, dp, ip, ip]
        L.pwo_max_dim.restype = ctypes.c_double
        _LIB = L
    return _LIB


_DP = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(_DP)


class Cage:
    """One molecule prepared for repeated point-vs-molecule evaluations."""

    def __init__(self, xyz, vdw, mass=None):
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        self.n = len(self.xyz)
        self.vdw = np.ascontiguousarray(vdw, dtype=np.float64)
        self.mass = None if mass is None else np.ascontiguousarray(mass, np.float64)
        self.xx = np.empty(self.n)
        _lib().pwo_row_sqnorms(self.n, _p(self.xyz), _p(self.xx))
        self.n_eval = 0

    def moved(self, xyz):
        return Cage(xyz, self.vdw, self.mass)

    def gap(self, point):
        """``min_i(|r_i - point| - vdw_i)`` and its first argmin.

        utilities.py:383-387 (``pore_diameter`` without the factor 2).
        """
        p = np.ascontiguousarray(point, dtype=np.float64)
        idx = ctypes.c_int64()
        v = _lib().pwo_min_gap(self.n, _p(self.xyz), _p(self.xx), _p(self.vdw), _p(p),
                               ctypes.byref(idx))
        self.n_eval += 1
        return v, idx.value


# --------------------------------------------------------------------------
# deterministic reductions
# --------------------------------------------------------------------------
def molecular_weight(mass) -> float:
    """utilities.py:96-107 -- numpy 1-D (pairwise) sum of the per-atom masses."""
    return float(np.array(list(mass)).sum())


def centre_of_mass(cage: Cage) -> np.ndarray:
    """utilities.py:127-148 -- column sums of x_ik*m_i over the total mass."""
    total = molecular_weight(cage.mass)
    weights = np.repeat(cage.mass[:, None], 3, axis=1)
    return np.sum(cage.xyz * weights, axis=0) / np.array([total, total, total])


def centroid(xyz) -> np.ndarray:
    """utilities.py:110-124."""
    return np.sum(xyz, axis=0) / xyz.shape[0]


def max_dim(cage: Cage):
    """utilities.py:355-372 -> (atom_1, atom_2, diameter)."""
    i = ctypes.c_int64()
    j = ctypes.c_int64()
    v = _lib().pwo_max_dim(cage.n, _p(cage.xyz), _p(cage.xx), _p(cage.vdw),
                           ctypes.byref(i), ctypes.byref(j))
    return i.value, j.value, float(v)


def pore_diameter(cage: Cage, centre=None):
    """utilities.py:375-388 -> (diameter, closest atom)."""
    if centre is None:
        centre = centre_of_mass(cage)
    g, i = cage.gap(centre)
    return float(g * 2), int(i)


def sphere_volume(r: float) -> float:
    """utilities.py:429-431."""
    return float(4 / 3 * np.pi * r**3)


def opt_pore_diameter(cage: Cage, trace=None):
    """utilities.py:400-426 -- L-BFGS-B from the COM inside the box COM +- r."""
    com = centre_of_mass(cage)
    r = pore_diameter(cage, com)[0] / 2
    box = tuple((com[k] - r, com[k] + r) for k in range(3))

    def neg_diameter(c):
        f = -(cage.gap(c)[0] * 2)
        if trace is not None:
            trace.append((np.array(c, float), f))
        return f

    res = minimize(neg_diameter, x0=com, bounds=box)
    d, a = pore_diameter(cage, res.x)
    return d, a, res.x, res


# --------------------------------------------------------------------------
# sampling sphere
# --------------------------------------------------------------------------
def n_sampling_points(radius: float, adjust: float = 1) -> int:
    """utilities.py:1399-1409 / :1606-1616."""
    return int(np.log10(4 * np.pi * radius**2) * 250 * adjust)


def sphere_points(radius: float, count: int) -> np.ndarray:
    """Golden-spiral points scaled to ``radius`` (utilities.py:1412-1423)."""
    golden = np.pi * (3 - np.sqrt(5))
    theta = golden * np.arange(count)
    z = np.linspace(1 - 1.0 / count, 1.0 / count - 1.0, count)
    ring = np.sqrt(1 - z * z)
    pts = np.zeros((count, 3))
    pts[:, 0] = ring * np.cos(theta) * radius
    pts[:, 1] = ring * np.sin(theta) * radius
    pts[:, 2] = z * radius
    return pts


def knn_eps(points: np.ndarray) -> float:
    """DBSCAN radius (utilities.py:1427-1434): m + sqrt(m), m = mean of all the
    10-nearest-neighbour distances, self-distance 0 included."""
    tree = KDTree(points)
    rows = []
    for p in points:
        dist, _ = tree.query(p.reshape(1, -1), k=10)
        rows.extend(dist)
    m = np.mean(rows)
    return float(m + m**0.5)


def _ray_hits(direction, xyz, vdw_col):
    """Forward intersections of the ray centroid -> direction with the atoms'
    vdW spheres (shared body of utilities.py:1138-1158 and :1561-1578).

    Returns a list of exit points p_1 for atoms that count as 'in the way'.
    """
    unit = direction / np.linalg.norm(direction)
    origin = centroid(xyz)
    rel = xyz - origin
    along = np.dot(rel, unit)
    with np.errstate(invalid="ignore"):
        perp = np.sqrt(np.einsum("ij,ij->i", rel, rel) - along**2)
        # (N,1) - (N,) broadcasts to N x N; only the diagonal is meaningful
        radicand = (vdw_col**2 - perp**2).diagonal()
    hits = []
    for k in np.argwhere(radicand > 0):
        half = np.sqrt(radicand[k[0]])
        t_in = along[k][0] - half
        t_out = along[k][0] + half
        p_in = origin + np.dot(t_in, unit)
        p_out = origin + np.dot(t_out, unit)
        if np.linalg.norm(p_in) < np.linalg.norm(p_out):
            hits.append(p_out)
    return hits


def path_scan(cage: Cage, vector, step: float):
    """utilities.py:1100-1129 -- walk origin -> vector in ``step`` increments.

    Returns ``[dist, 2*gap, p(3), vector(3)]`` at the narrowest point if every
    sampled point is outside all vdW spheres, else ``None``.
    """
    pieces = int(np.linalg.norm(vector) // step)
    hop = vector / pieces
    gaps = np.array([cage.gap(hop * k)[0] for k in range(pieces + 1)])
    if all(g > 0 for g in gaps):
        k = int(np.argmin(gaps))
        return np.array([np.linalg.norm(hop * k), gaps[k] * 2, *(hop * k), *vector])
    return None


def find_average_diameter(cage: Cage, adjust: float = 1) -> float:
    """utilities.py:1586-1650."""
    com = centre_of_mass(cage)
    shifted = cage.moved(cage.xyz - np.array([com] * cage.n))
    radius = max_dim(shifted)[2]
    pts = sphere_points(radius, n_sampling_points(radius, adjust))
    vdw_col = shifted.vdw.reshape(-1, 1)
    far = []
    for p in pts:
        exits = _ray_hits(p, shifted.xyz, vdw_col)
        if exits:
            ranked = sorted(([np.linalg.norm(e), e] for e in exits), reverse=True,
                            key=lambda t: t[0])
            far.append(float(np.linalg.norm(ranked[0][1])))
    return float(np.mean(far) * 2)


# --------------------------------------------------------------------------
# windows
# --------------------------------------------------------------------------
def _angle(u, v):
    """utilities.py:1088-1097 (arccos of the *absolute* cosine)."""
    c = abs(u[0] * v[0] + u[1] * v[1] + u[2] * v[2]) / (
        np.sqrt(u[0] ** 2 + u[1] ** 2 + u[2] ** 2) * np.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2)
    )
    return np.arccos(c)


def _octant_angles(v):
    """Rotation angles taking ``v`` onto +z: utilities.py:1235-1259."""
    a1 = _angle(np.array([v[0], v[1], 0]), np.array([1, 0, 0]))
    a2 = _angle(v, np.array([0, 0, 1]))
    sx, sy, sz = v[0] >= 0, v[1] >= 0, v[2] >= 0
    # the reference applies eight independent ifs in sequence; they are mutually
    # exclusive, so a lookup is equivalent
    if sz:
        if sx and sy:
            a1, a2 = -a1, -a2
        elif (not sx) and sy:
            a1 = np.pi * 2 + a1
        elif sx and (not sy):
            a2 = -a2
        else:
            a1 = np.pi * 2 - a1
    else:
        if sx and sy:
            a1, a2 = -a1, np.pi + a2
        elif (not sx) and sy:
            a2 = np.pi - a2
        elif sx and (not sy):
            a2 = a2 + np.pi
        else:
            a1, a2 = -a1, np.pi - a2
    return a1, a2


def _rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])


def _rot_y(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])


def window_fit(cage: Cage, cluster_rows: np.ndarray, detail=None, increment2: float = 0.1,
               z_bounds=None, lb_z: bool = True, z_second_mini: bool = False):
    """utilities.py:1191-1361 -- measure one window from its cluster of vectors."""
    z_bounds = [None, None] if z_bounds is None else list(z_bounds)
    best = cluster_rows[cluster_rows.argmax(axis=0)[1]][5:8]
    fine = path_scan(cage, best, increment2)
    if fine is None:
        return None
    v = fine[5:8]
    a1, a2 = _octant_angles(v)
    xyz = np.array([np.dot(_rot_z(a1), r) for r in cage.xyz])
    xyz = np.array([np.dot(_rot_y(a2), r) for r in xyz])
    neck = fine[0]
    xyz = xyz - np.array([[0, 0, neck]] * cage.n)
    local = cage.moved(xyz)
    centre = np.array([0, 0, 0], dtype=float)
    d0 = pore_diameter(local, centre)[0]

    def along_z(z):
        return local.gap(np.array([centre[0], centre[1], z[0]]))[0] * 2

    if lb_z:
        z_bounds[0] = -neck
    zres = minimize(along_z, x0=centre[2], bounds=[z_bounds])
    centre[2] = zres.x[0]

    def in_plane(xy):
        return -(local.gap(np.array([xy[0], xy[1], centre[2]]))[0] * 2)

    box = ((-d0 / 2, d0 / 2), (-d0 / 2, d0 / 2))
    xyres = brute(in_plane, box, full_output=True, finish=fmin)
    centre[0] = xyres[0][0]
    centre[1] = xyres[0][1]
    if z_second_mini is not False:
        zres = minimize(along_z, x0=centre[2], bounds=[z_bounds])
        centre[2] = zres.x[0]
    diameter = pore_diameter(local, centre)[0]
    centre[2] = centre[2] + neck
    centre = np.dot(_rot_y(-a2), centre)
    centre = np.dot(_rot_z(-a1), centre)
    if detail is not None:
        detail.append({"vector": v, "fine": fine, "a1": a1, "a2": a2, "z": zres,
                       "xy": xyres[0], "rot": xyz})
    return diameter, centre


def find_windows(cage: Cage, detail=None, adjust: float = 1, pore_opt: bool = True,
                 increment: float = 1.0, **window_options):
    """utilities.py:1364-1553 (``adjust``, ``pore_opt``, ``increment`` as there);
    ``window_options`` are window_analysis's keywords (increment2, z_bounds, lb_z,
    z_second_mini), which the reference's find_windows leaves at their defaults.

    Returns ``None`` (no window), or ``(diameters (W,), centres (W,3))``.
    """
    com = centre_of_mass(cage)
    if pore_opt is True:
        pore_centre = opt_pore_diameter(cage)[2]
        shift = com - pore_centre
        origin_back = com - shift
    else:
        shift = np.zeros(3)
        origin_back = com
    # shift_com(elements, coordinates, com_adjust): coords - (COM - adjust)
    moved = cage.moved(cage.xyz - np.array([com - shift] * cage.n))
    radius = max_dim(moved)[2] / 2
    count = n_sampling_points(radius, adjust)
    pts = sphere_points(radius, count)
    eps = knn_eps(pts)
    vdw_col = moved.vdw.reshape(-1, 1)
    rows = []
    kept = []
    for k, p in enumerate(pts):
        if len(_ray_hits(p, moved.xyz, vdw_col)) == 0:
            r = path_scan(moved, p, increment)
            if r is not None:
                rows.append(r)
                kept.append(k)
    if detail is not None:
        detail.update({"radius": radius, "count": count, "eps": eps, "kept": kept,
                       "rows": np.array(rows), "windows": []})
    if not rows:
        return None
    ends = np.array([r[5:8] for r in rows])
    labels = DBSCAN(eps=eps).fit(ends).labels_
    if detail is not None:
        detail["labels"] = np.array(labels)
    groups = {}
    for lab in set(labels):
        groups[lab] = []
    for r, lab in zip(rows, labels):
        groups[lab].append(r)
    found = []
    for lab in groups:
        if lab == -1:
            continue
        found.append(window_fit(moved, np.array(groups[lab]),
                                None if detail is None else detail["windows"], **window_options))
    diam = np.array([w[0] for w in found if w is not None])
    cen = np.array([np.add(w[1], origin_back) for w in found if w is not None])
    return diam, cen


# --------------------------------------------------------------------------
# the unit of work
# --------------------------------------------------------------------------
def full_analysis(xyz, vdw, mass) -> dict:
    """One (frame, molecule) unit: molecular.py:156-202, de-duplicated.

    Returns the flat record layout used by the golden fixtures and by the HIP
    library's ``pw_unit_out`` (include/pywindow_amd.h).
    """
    cage = Cage(xyz, vdw, mass)
    out = {"n_atoms": cage.n, "mw": molecular_weight(cage.mass)}
    out["com"] = centre_of_mass(cage)
    i, j, d = max_dim(cage)
    out.update(maxd=d, maxd_i=i, maxd_j=j)
    out["avg_d"] = find_average_diameter(cage)
    d, a = pore_diameter(cage)
    out.update(pore_d=d, pore_atom=a, pore_vol=sphere_volume(d / 2))
    d, a, c, _ = opt_pore_diameter(cage)
    out.update(pore_opt_d=d, pore_opt_atom=a, pore_opt_c=np.array(c),
               pore_vol_opt=sphere_volume(d / 2))
    win = find_windows(cage)
    wd = np.full(W_MAX, np.nan)
    wc = np.full((W_MAX, 3), np.nan)
    if win is None:
        out["n_windows"] = -1
    else:
        n = len(win[0])
        out["n_windows"] = n
        wd[:n] = win[0]
        if n:
            wc[:n] = win[1]
    out.update(win_d=wd, win_c=wc)
    out["n_eval"] = cage.n_eval
    return out
