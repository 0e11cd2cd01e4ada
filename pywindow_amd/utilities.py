"""Function-level mirror of the reference's hot-path interface.

Same names, argument meaning and error behaviour as the free functions
``molecular.py:29-44`` imports from the reference's ``utilities.py``; every call
is executed by the HIP engine (a batch of one molecule).  For bulk work use
:mod:`pywindow_amd.trajectory` / :func:`pywindow_amd.engine.analyse`, which put
all molecules x frames into one launch.
"""

from __future__ import annotations

import numpy as np

from . import _lib, engine
from .element_data import MASS, VDW, atomic_covalent_radius, atomic_mass, atomic_vdw_radius, element_ids  # noqa: F401  (re-exported)
from .rebuild import discrete_molecules, lattice_array_to_unit_cell, unit_cell_to_lattice_array  # noqa: F401


def _one(elements, coordinates, stages, params=None):
    return engine.analyse([(elements, coordinates)], stages, params=params)[0]


def molecular_weight(elements) -> float:
    """Reference utilities.py:96-107."""
    dummy = np.zeros((len(elements), 3))
    return float(_one(elements, dummy + np.arange(len(elements))[:, None], _lib.STAGE_BASIC)["mw"])


def center_of_mass(elements, coordinates) -> np.ndarray:
    """Reference utilities.py:127-148."""
    return np.array(_one(elements, coordinates, _lib.STAGE_BASIC)["com"])


def shift_com(elements, coordinates, com_adjust=np.zeros(3)) -> np.ndarray:  # noqa: B008
    """Reference utilities.py:344-352 (coordinates translated by COM - adjust)."""
    com = center_of_mass(elements, coordinates)
    coordinates = np.asarray(coordinates, dtype=float)
    return coordinates - np.array([com - com_adjust] * coordinates.shape[0])


def max_dim(elements, coordinates) -> tuple[int, int, float]:
    """Reference utilities.py:355-372."""
    r = _one(elements, coordinates, _lib.STAGE_BASIC)
    return int(r["maxd_i"]), int(r["maxd_j"]), float(r["maxd"])


def pore_diameter(elements, coordinates, com=None) -> tuple[float, int]:
    """Reference utilities.py:375-388."""
    if com is None:
        r = _one(elements, coordinates, _lib.STAGE_BASIC)
        return float(r["pore_d"]), int(r["pore_atom"])
    gaps, idx = engine.context().point_gaps(
        engine.make_batch([(elements, coordinates)]), np.zeros(1, np.int64), np.asarray(com, float).reshape(1, 3)
    )
    return float(gaps[0] * 2), int(idx[0])


def sphere_volume(sphere_radius: float) -> float:
    """Reference utilities.py:429-431."""
    return float(4 / 3 * np.pi * sphere_radius**3)


def opt_pore_diameter(elements, coordinates, bounds=None, com=None):
    """Reference utilities.py:400-426.  ``com``: start of the optimisation (default the centre
    of mass); ``bounds``: three ``(lo, hi)`` pairs as for ``scipy.optimize.minimize`` (default:
    start -/+ the pore radius at the start)."""
    params = None
    if bounds is not None or com is not None:
        if bounds is not None:
            for lo, hi in bounds:
                if lo is not None and hi is not None and lo > hi:
                    raise ValueError("An upper bound is less than the corresponding lower bound.")
        params = _lib.Params(opt_start=com, opt_bounds=bounds)
    r = _one(elements, coordinates, _lib.STAGE_OPT, params)
    engine.raise_like_reference(r)      # non-porous molecule: scipy's ValueError (inverted default box)
    return float(r["pore_opt_d"]), int(r["pore_opt_atom"]), np.array(r["pore_opt_c"])


def find_average_diameter(elements, coordinates, adjust=1, processes=None) -> float:
    """Reference utilities.py:1586-1650; ``adjust`` scales the number of sampling rays.
    ``processes`` is accepted and ignored (there is no CPU pool)."""
    del processes
    params = None if adjust == 1 else _lib.Params(adjust_average=adjust)
    r = _one(elements, coordinates, _lib.STAGE_AVG, params)
    engine.raise_on_capacity(r)
    return float(r["avg_d"])


def find_windows(elements, coordinates, processes=None, adjust=1, pore_opt=True, increment=1.0,
                 increment2=0.1, z_bounds=None, lb_z=True, z_second_mini=False):
    """Reference utilities.py:1364-1553.  ``adjust`` scales the number of sampling vectors,
    ``pore_opt`` centres the molecule on the optimised pore (``is True``, like the reference)
    or on its centre of mass, ``increment`` is the step of the coarse path scan.

    The last four keywords are those of the reference's ``window_analysis``
    (utilities.py:1191-1200), which its ``find_windows`` always calls with the defaults: the
    refined path-scan step, the bounds of the neck search along the window axis (``lb_z``: the
    lower bound is the distance back to the pore centre), and the optional second neck search.
    """
    del processes
    default = (adjust == 1 and pore_opt is True and increment == 1.0 and increment2 == 0.1
               and z_bounds is None and lb_z is True and z_second_mini is False)
    params = None if default else _lib.Params(
        adjust_windows=adjust, pore_opt=pore_opt is True, increment=increment, increment2=increment2,
        z_bounds=z_bounds, lb_z=bool(lb_z), z_second_mini=z_second_mini)
    extra: list = []
    r = engine.analyse([(elements, coordinates)], _lib.STAGE_WINDOWS, params=params, extra=extra)[0]
    engine.raise_on_capacity(r)
    if int(r["status"]) & _lib.ST_TOO_FEW_POINTS:
        # sklearn's KDTree.query(k=10) raises this inside the reference (utilities.py:1428-1431)
        raise ValueError("k must be less than or equal to the number of training points")
    if int(r["status"]) & _lib.ST_Z_BOUNDS:
        # scipy.optimize.minimize raises this from inside the reference's window_analysis
        raise ValueError("An upper bound is less than the corresponding lower bound.")
    if pore_opt is True:
        engine.raise_like_reference(r)      # (without the pore optimisation there are no bounds to invert)
        engine.warn_like_reference(r)
    else:
        engine.warn_like_reference(np.array(int(r["status"]) & ~_lib.ST_NEGATIVE_PORE, dtype=[("status", np.int32)]))
    return engine.windows_of(r, engine.extra_by_unit(extra).get(0))


# ---- shape descriptors and circumcircles (reference utilities.py:434-650, 1653-1691) ---------------
def _shape(elements, coordinates):
    ids = element_ids(elements)
    xyz = np.ascontiguousarray(coordinates, dtype=np.float64)
    batch = _lib.Batch(np.array([0, len(xyz)], dtype=np.int64), xyz, VDW[ids], MASS[ids])
    return engine.context().shape(batch)[0]


def get_gyration_tensor(elements, coordinates) -> np.ndarray:
    """Reference utilities.py:461-495 (bit-identical)."""
    return np.array(_shape(elements, coordinates)["gyration"])


def get_inertia_tensor(elements, coordinates) -> np.ndarray:
    """Reference utilities.py:498-529, including its (N, 1) x (N,) broadcast (bit-identical)."""
    return np.array(_shape(elements, coordinates)["inertia"])


def calc_asphericity(elements, coordinates) -> float:
    """Reference utilities.py:626-632 (eigenvalues: Jacobi instead of LAPACK, a few ulps)."""
    return float(_shape(elements, coordinates)["asphericity"])


def calc_acylidricity(elements, coordinates) -> float:
    """Reference utilities.py:635-641."""
    return float(_shape(elements, coordinates)["acylidricity"])


def calc_relative_shape_anisotropy(elements, coordinates) -> float:
    """Reference utilities.py:644-650."""
    return float(_shape(elements, coordinates)["relative_shape_anisotropy"])


def inertia_eigenvalues(elements, coordinates) -> np.ndarray:
    """``get_tensor_eigenvalues(get_inertia_tensor(...), sort=True)`` (utilities.py:449-458)."""
    return np.array(_shape(elements, coordinates)["eigenvalues"])


def circumcircle_window(coordinates, atom_set):
    """Reference utilities.py:1653-1676: ``(radius, centre)`` of the circle through three atoms,
    less a carbon van der Waals radius."""
    d, c = engine.context().circumcircle(coordinates, [list(atom_set)[:3]])
    return float(d[0]) / 2, c[0]


def circumcircle(coordinates, atom_sets):
    """Reference utilities.py:1679-1691: ``(diameters, centres)`` lists for atom triples."""
    sets = [[int(i) for i in list(t)[:3]] for t in atom_sets]
    if not sets:
        return [], []
    d, c = engine.context().circumcircle(coordinates, sets)
    return [float(x) for x in d], [np.array(x) for x in c]


# ---- principal axes (reference utilities.py:532-623; only used by Molecule._align_to_principal_axes) ----
def principal_axes(elements, coordinates) -> np.ndarray:
    """Reference utilities.py:532-536: the eigenvectors of the inertia tensor, one per ROW.

    The tensor (the O(N^2) part, with the reference's accidental N x N broadcast) comes from the GPU
    bit for bit; the 3 x 3 eigen-decomposition is the same ``numpy.linalg.eig`` call the reference
    makes, so order and signs are LAPACK ``dgeev``'s, as there: eigenvalues in the order the QR
    iteration deflates them (NOT sorted), every vector of unit length with the sign ``dgeev`` leaves
    it with.  Pinned by tests/golden/axes.npz.
    """
    return np.linalg.eig(get_inertia_tensor(elements, coordinates))[1].T


def normalize_vector(vector) -> np.ndarray:
    """Reference utilities.py:539-556: unit vector, ROUNDED to four decimals (so the rotation below is
    about a slightly different axis than the one asked for -- reproduced)."""
    return np.round(np.divide(vector, np.linalg.norm(vector)), decimals=4)


def rotation_matrix_arbitrary_axis(angle, axis) -> np.ndarray:
    """Reference utilities.py:559-591: rotation by ``angle`` about ``axis`` from the unit quaternion
    (w, x, y, z) = (cos(angle/2), sin(angle/2) * axis / |axis|), sums taken left to right as there."""
    w = np.cos(angle / 2)
    x, y, z = normalize_vector(axis) * np.sin(angle / 2)
    ww, xx, yy, zz = np.square(w), np.square(x), np.square(y), np.square(z)
    return np.array([
        [ww + xx - yy - zz, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), ww + yy - xx - zz, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), ww + zz - xx - yy],
    ])


def align_principal_ax(elements, coordinates):
    """Reference utilities.py:594-623: three successive rotations that turn principal axes 2, 1, 0
    onto x, y, z.  Returns ``(coordinates, [rotation matrices])``.  As in the reference the axes are
    those of the ORIGINAL coordinates in all three steps (it never recomputes them from the rotated
    copy), and every atom is rotated by its own 3 x 3 by 3 x 1 product."""
    moved = np.array(coordinates, dtype=float)
    rotations = []
    axes = principal_axes(elements, coordinates)        # (the reference recomputes the same thing three times)
    for which, target in ((2, (1, 0, 0)), (1, (0, 1, 0)), (0, (0, 0, 1))):
        target = np.array(target)
        normal = np.cross(axes[which], target)
        angle = np.arctan2(np.linalg.norm(normal), np.dot(axes[which], target))
        rot = np.matrix(rotation_matrix_arbitrary_axis(angle, normal))
        rotations.append(rot)
        moved = np.array([np.array((rot * row.reshape(-1, 1)).reshape(1, -1))[0] for row in moved])
    return moved, rotations
