"""CPU oracle for the shape descriptors and the circumcircle window estimate
(SURVEY.md 8f-4).  TEST INFRASTRUCTURE: only ``tests/`` may import it.

Restates utilities.py:434-650 and :1653-1691 (paths relative to
``/root/reference/src/pywindow/_internal/``) with the reference's own numpy
expressions -- including the (N, 1) x (N,) broadcast in the inertia tensor, which makes
every entry a sum over all N x N (mass_i, position_j) pairs.

PARITY PIN: tests/golden/shape.npz, produced by the reference in the development
container (tests/golden/make_golden.py shape); see tests/test_shape.py.
"""
from __future__ import annotations

import numpy as np


def gyration_tensor(xyz, mass) -> np.ndarray:
    """utilities.py:461-495: coordinates relative to the centre of mass, second moments / N."""
    xyz = np.asarray(xyz, float)
    mass = np.asarray(mass, float)
    com = np.sum(xyz * mass.reshape(-1, 1), axis=0) / np.sum(mass)      # utilities.py:127-148
    c = xyz - com
    diag = np.sum(c ** 2, axis=0)
    xy = np.sum(c[:, 0] * c[:, 1])
    xz = np.sum(c[:, 0] * c[:, 2])
    yz = np.sum(c[:, 1] * c[:, 2])
    return np.array([[diag[0], xy, xz], [xy, diag[1], yz], [xz, yz, diag[2]]]) / c.shape[0]


def inertia_tensor(xyz, mass) -> np.ndarray:
    """utilities.py:498-529: the mass column (N, 1) against (N,) rows broadcasts to (N, N)."""
    xyz = np.asarray(xyz, float)
    m = np.asarray(mass, float).reshape(-1, 1)
    p2 = xyz ** 2
    d1 = np.sum(m * (p2[:, 1] + p2[:, 2]))
    d2 = np.sum(m * (p2[:, 0] + p2[:, 2]))
    d3 = np.sum(m * (p2[:, 0] + p2[:, 1]))
    mxy = np.sum(-m * xyz[:, 0] * xyz[:, 1])
    mxz = np.sum(-m * xyz[:, 0] * xyz[:, 2])
    myz = np.sum(-m * xyz[:, 1] * xyz[:, 2])
    return np.array([[d1, mxy, mxz], [mxy, d2, myz], [mxz, myz, d3]]) / xyz.shape[0]


def sorted_eigenvalues(tensor) -> np.ndarray:
    """utilities.py:449-458 with sort=True."""
    return np.array(sorted(np.linalg.eigvals(tensor), reverse=True), dtype=np.float64)


def descriptors(eig):
    """asphericity, acylidricity, relative shape anisotropy: utilities.py:434-446."""
    a = eig[0] - (eig[1] + eig[2]) / 2
    b = eig[1] - eig[2]
    k = 1 - 3 * ((eig[0] * eig[1] + eig[0] * eig[2] + eig[1] * eig[2]) / (np.sum(eig)) ** 2)
    return np.array([a, b, k])


def circumcircle_window(xyz, atom_set):
    """utilities.py:1653-1676: circumscribed circle of three atoms, minus a carbon radius."""
    pa, pb, pc = (np.array(xyz[int(i)]) for i in atom_set)
    a = np.linalg.norm(pc - pb)
    b = np.linalg.norm(pc - pa)
    c = np.linalg.norm(pb - pa)
    s = (a + b + c) / 2
    r = a * b * c / 4 / np.sqrt(s * (s - a) * (s - b) * (s - c)) - 1.70
    b1 = a * a * (b * b + c * c - a * a)
    b2 = b * b * (a * a + c * c - b * b)
    b3 = c * c * (a * a + b * b - c * c)
    com = np.column_stack((pa, pb, pc)).dot(np.hstack((b1, b2, b3)))
    com /= b1 + b2 + b3
    return r, com


def circumcircle(xyz, atom_sets):
    """utilities.py:1679-1691: diameters and centres for a list of atom triples."""
    out = [circumcircle_window(xyz, t) for t in atom_sets]
    return [2 * r for r, _ in out], [c for _, c in out]
