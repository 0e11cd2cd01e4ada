"""CPU restatement of the reference's periodic pre-processing -- TEST INFRASTRUCTURE.

SURVEY.md section 8 row f-1: ``create_supercell`` (utilities.py:768-810) and
``discrete_molecules`` (utilities.py:820-1085) as driven by
``MolecularSystem.rebuild_system`` / ``make_modular`` (molecular.py:672-708,
798-824).  Only ``tests/`` may import this module; it is the checker for the HIP
implementation in ``pywindow_amd/csrc/pw_rebuild.hpp``, never the product path.

The reference keeps atoms as Python lists ``[element, atom_id, x, y, z]`` (coordinates
rounded to 8 decimals, utilities.py:187-220) and compares / removes them *by value*.
This restatement keeps that semantics but works on integer keys: every distinct value
gets one key, the ordered ``atom_list`` becomes an ordered set of remaining unit-cell
indices, and ``x in atom_list`` / ``atom_list.remove(x)`` / ``unique`` /
``x not in final_molecule`` become key look-ups.  Distances use the same two
formulas as the reference: scikit-learn's ``euclidean_distances`` (N x 1 call shape,
C primitive) for the pre-filter and start atom, ``distance()`` (utilities.py:80-93)
for the bond test.

Pinned against the reference itself by tests/golden/make_golden.py (group
``rebuild``) and tests/test_rebuild.py.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import pw_oracle as O

TERMINAL = ("H", "CL", "BR", "F", "HE", "AR", "NE", "KR", "XE", "RN")   # utilities.py:943

_DP = ctypes.POINTER(ctypes.c_double)


def _lib():
    L = O._lib()
    if not getattr(L, "_rebuild_ready", False):
        L.pwo_dists.argtypes = [ctypes.c_int64, _DP, _DP, _DP, _DP]
        L.pwo_mat3_apply.argtypes = [_DP, ctypes.c_int64, _DP, _DP]
        L.pwo_round8.argtypes = [ctypes.c_int64, _DP, _DP]
        L._rebuild_ready = True
    return L


def _p(a):
    return a.ctypes.data_as(_DP)


# ---- lattice helpers (utilities.py:653-709) ---------------------------------------------------
def unit_cell_to_lattice_array(cryst) -> np.ndarray:
    a_, b_, c_, alpha, beta, gamma = cryst
    ra, rb, rg = np.deg2rad(alpha), np.deg2rad(beta), np.deg2rad(gamma)
    volume = a_ * b_ * c_ * (
        1 - np.cos(ra) ** 2 - np.cos(rb) ** 2 - np.cos(rg) ** 2 + 2 * np.cos(ra) * np.cos(rb) * np.cos(rg)
    ) ** 0.5
    return np.array([
        [a_, b_ * np.cos(rg), c_ * np.cos(rb)],
        [0, b_ * np.sin(rg), c_ * (np.cos(ra) - np.cos(rb) * np.cos(rg)) / np.sin(rg)],
        [0, 0, volume / (a_ * b_ * np.sin(rg))],
    ])


def lattice_array_to_unit_cell(lattice) -> np.ndarray:
    lengths = np.sqrt(np.sum(lattice ** 2, axis=0))
    gamma_r = np.arccos(lattice[0][1] / lengths[1])
    beta_r = np.arccos(lattice[0][2] / lengths[2])
    alpha_r = np.arccos(lattice[1][2] * np.sin(gamma_r) / lengths[2] + np.cos(beta_r) * np.cos(gamma_r))
    return np.append(lengths, [np.rad2deg(alpha_r), np.rad2deg(beta_r), np.rad2deg(gamma_r)])


def mat3_apply(matrix, points) -> np.ndarray:
    """Rows of ``points`` through ``np.matrix(matrix) * row.reshape(-1, 1)``
    (utilities.py:722-743), bit for bit."""
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    x = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    y = np.empty_like(x)
    _lib().pwo_mat3_apply(_p(m), len(x), _p(x), _p(y))
    return y


def round8(values) -> np.ndarray:
    """Python ``round(float(x), 8)`` element-wise (compose_atom_list, utilities.py:187-220)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty_like(v)
    _lib().pwo_round8(v.size, _p(v), _p(out))
    return out


def create_supercell(system: dict, supercell=None) -> dict:
    """utilities.py:768-810 -- 3x3x3 images, image loops a, b, c nested in that order."""
    if supercell is None:
        supercell = [[-1, 1], [-1, 1], [-1, 1]]
    matrix = system["lattice"] if "lattice" in system else unit_cell_to_lattice_array(system["unit_cell"])
    coords = np.asarray(system["coordinates"], dtype=np.float64)
    frac = mat3_apply(np.linalg.inv(matrix), coords)
    blocks = []
    for a_ in range(supercell[0][0], supercell[0][1] + 1):
        for b_ in range(supercell[1][0], supercell[1][1] + 1):
            for c_ in range(supercell[2][0], supercell[2][1] + 1):
                blocks.append(frac + np.array([[a_, b_, c_]]))
    all_frac = np.concatenate(blocks, axis=0)
    out = {
        "elements": np.concatenate([system["elements"]] * len(blocks)),
        "coordinates": mat3_apply(matrix, all_frac),
        "unit_cell": lattice_array_to_unit_cell(matrix),
        "lattice": matrix,
    }
    if "atom_ids" in system:
        out["atom_ids"] = np.concatenate([system["atom_ids"]] * len(blocks))
    return out


def _centre_of_mass(elements, coordinates) -> np.ndarray:
    from pywindow_amd import element_data as E

    mass = E.MASS[E.element_ids(elements)]
    return O.centre_of_mass(O.Cage(np.asarray(coordinates, float), np.zeros(len(mass)), mass))


class _Atoms:
    """Value-keyed view of an atom list."""

    def __init__(self, elements, ids, coords):
        from pywindow_amd import element_data as E

        self.elements = np.asarray(elements)
        self.ids = None if ids is None else np.asarray(ids)
        self.xyz = round8(np.asarray(coords, dtype=np.float64)).reshape(-1, 3)
        self.n = len(self.xyz)
        upper = [str(e).upper() for e in self.elements]
        self.heavy = np.array([u not in TERMINAL for u in upper], dtype=bool)
        self.rcov = np.array([E.atomic_covalent_radius[u] for u in upper])
        self.xx = np.empty(self.n)
        O._lib().pwo_row_sqnorms(self.n, _p(self.xyz), _p(self.xx))

    def value(self, k):
        base = (str(self.elements[k]),) if self.ids is None else (str(self.elements[k]), str(self.ids[k]))
        return base + tuple(self.xyz[k].tolist())

    def dists(self, point) -> np.ndarray:
        out = np.empty(self.n)
        p = np.ascontiguousarray(point, dtype=np.float64)
        _lib().pwo_dists(self.n, _p(self.xyz), _p(self.xx), _p(p), _p(out))
        return out


def _bond_length(a, b) -> np.ndarray:
    """``distance()`` of utilities.py:80-93 for rows of ``b`` against point ``a``."""
    d = a[None, :] - b
    sq = d ** 2
    return ((sq[:, 0] + sq[:, 1]) + sq[:, 2]) ** 0.5


def discrete_molecules(system: dict, rebuild: dict | None = None, tol: float = 0.4) -> list[dict]:
    """utilities.py:820-1085."""
    from pywindow_amd import element_data as E

    if rebuild is not None:
        mode = 3
    elif "unit_cell" in system:
        mode = 2 if system["unit_cell"].shape == (6,) else 1
    elif "lattice" in system:
        mode = 2 if system["lattice"].shape == (3, 3) else 1
    else:
        mode = 1
    elements = system["elements"]
    coordinates = np.asarray(system["coordinates"], dtype=np.float64)
    has_ids = "atom_ids" in system
    cell = _Atoms(elements, system["atom_ids"] if has_ids else None, coordinates)
    matrix = None
    boundary = None
    if mode in (2, 3):
        origin = np.array([0.01, 0.0, 0.0])
        matrix = system["lattice"] if "lattice" in system else unit_cell_to_lattice_array(system["unit_cell"])
        pseudo_origin = mat3_apply(matrix, np.array([0.26, 0.25, 0.25]))[0]
        system_com = _centre_of_mass(elements, coordinates)
        boundary = np.array([-0.5, 0.5]) if np.allclose(system_com, origin, atol=1e-00) else np.array([0.0, 1.0])
    else:
        pseudo_origin = _centre_of_mass(elements, coordinates) + np.array([0.01, 0.0, 0.0])
    sup = None
    if rebuild is not None:
        # utilities.py:898-902: the supercell atom list always carries atom ids
        sup = _Atoms(rebuild["elements"], rebuild["atom_ids"], rebuild["coordinates"])
    # value keys: one integer per distinct [element, (id,) x, y, z]
    keys: dict = {}
    info: list = []          # key -> (which table, index): representative for element / coords / id

    def key_of(table, idx, tag):
        v = table.value(idx)
        k = keys.get(v)
        if k is None:
            k = len(info)
            keys[v] = k
            info.append((tag, idx))
        return k

    cell_key = np.array([key_of(cell, k, 0) for k in range(cell.n)])
    sup_key = None
    if sup is not None:
        if not has_ids:
            # the reference compares a 5-item supercell entry with 4-item cell entries: never equal
            sup_key = np.array([len(info) + k for k in range(sup.n)])
            info.extend((1, k) for k in range(sup.n))
        else:
            sup_key = np.array([key_of(sup, k, 1) for k in range(sup.n)])
    remaining = np.ones(cell.n, dtype=bool)           # atom_list membership, in order
    left_of_key: dict = {}
    for k in cell_key:
        left_of_key[k] = left_of_key.get(k, 0) + 1
    max_r_cov = max(E.atomic_covalent_radius[str(e).upper()] for e in set(system["elements"]))
    max_dist = 2 * max_r_cov + tol
    molecules = []

    def table_of(key):
        tag, idx = info[key]
        return (cell if tag == 0 else sup), idx

    while remaining.any():
        heavy_left = np.nonzero(remaining & cell.heavy)[0]
        if len(heavy_left) == 0:
            break
        d0 = cell.dists(pseudo_origin)[heavy_left]
        working = [int(cell_key[heavy_left[int(np.argmin(d0))]])]
        final: list[int] = []
        in_final: set[int] = set()
        while working:
            found: list[int] = []
            left_idx = np.nonzero(remaining)[0]
            for key in working:
                tab, idx = table_of(key)
                if tab.heavy[idx]:
                    here = tab.xyz[idx]
                    r_i = tab.rcov[idx]
                    if len(left_idx):
                        d = cell.dists(here)[left_idx]
                        near = left_idx[(d > 0.1) * (d < max_dist)]
                        if len(near):
                            r = _bond_length(here, cell.xyz[near])
                            rc = r_i + cell.rcov[near]
                            found.extend(int(q) for q in cell_key[near[(rc - tol < r) & (r < rc + tol)]])
                    if sup is not None:
                        d = sup.dists(here)
                        near = np.nonzero((d > 0.1) * (d < max_dist))[0]
                        near = np.array([s for s in near if left_of_key.get(int(sup_key[s]), 0) == 0], dtype=int)
                        if len(near):
                            r = _bond_length(here, sup.xyz[near])
                            rc = r_i + sup.rcov[near]
                            found.extend(int(q) for q in sup_key[near[(rc - tol < r) & (r < rc + tol)]])
                final.append(key)
                in_final.add(key)
            for key in working:            # atom_list.remove(i): the first remaining entry of equal value
                if left_of_key.get(key, 0) > 0:
                    first = np.nonzero(remaining & (cell_key == key))[0][0]
                    remaining[first] = False
                    left_of_key[key] -= 1
            working = []
            seen: set[int] = set()
            for key in found:
                if key not in seen:
                    seen.add(key)
                    if key not in in_final:
                        working.append(key)
        tabs = [table_of(k) for k in final]
        mol = {
            "elements": np.array([str(t.elements[i]) for t, i in tabs], dtype="str"),
            "coordinates": np.array([t.xyz[i] for t, i in tabs]).reshape(-1, 3),
        }
        if has_ids:
            mol["atom_ids"] = np.array([str(t.ids[i]) for t, i in tabs], dtype="str")
        keep = True
        if rebuild is not None:
            com = _centre_of_mass(mol["elements"], mol["coordinates"])
            com_frac = mat3_apply(np.linalg.inv(matrix), com)[0]
            rounded = np.around(com_frac, decimals=8)
            keep = bool(np.all(np.logical_and(rounded >= boundary[0], rounded < boundary[1]), axis=0))
        if keep:
            molecules.append(mol)
    return molecules


def rebuild_system(system: dict) -> dict:
    """molecular.py:672-708 -- concatenation of the rebuilt molecules."""
    parts = discrete_molecules(system, rebuild=create_supercell(system))
    return {
        "coordinates": np.concatenate([np.zeros((0, 3))] + [m["coordinates"] for m in parts], axis=0),
        "atom_ids": np.concatenate([np.array([])] + [m["atom_ids"] for m in parts], axis=0),
        "elements": np.concatenate([np.array([])] + [m["elements"] for m in parts], axis=0),
    }


def make_modular(system: dict, rebuild: bool = False) -> list[dict]:
    """molecular.py:798-824."""
    return discrete_molecules(system, rebuild=create_supercell(system) if rebuild is True else None)
