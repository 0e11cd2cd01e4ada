"""Host side of the periodic pre-processing: marshals a molecular
system for ``pw_discrete_molecules`` (csrc/pw_rebuild.hpp) and turns its output back
into the reference's list of molecule dicts.

Counterpart of ``create_supercell`` / ``discrete_molecules`` (reference
utilities.py:768-810, 820-1085).  Only per-system constants are computed here, with
the same numpy calls the reference makes (lattice matrix, its inverse); every
per-atom operation runs in the HIP kernel.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .element_data import COVALENT, MASS, TERMINAL_SYMBOLS, atomic_covalent_radius, element_ids

ST_NB_OVERFLOW = 1
ST_SEG_OVERFLOW = 2
ST_ATOMS_OVERFLOW = 4
ST_MOLS_OVERFLOW = 8
ST_THIN_CELL = 16


def unit_cell_to_lattice_array(cryst) -> np.ndarray:
    """Reference utilities.py:653-690 (same numpy expression, evaluated on the host)."""
    a_, b_, c_, alpha, beta, gamma = cryst
    r_alpha, r_beta, r_gamma = np.deg2rad(alpha), np.deg2rad(beta), np.deg2rad(gamma)
    volume = a_ * b_ * c_ * (
        1 - np.cos(r_alpha) ** 2 - np.cos(r_beta) ** 2 - np.cos(r_gamma) ** 2
        + 2 * np.cos(r_alpha) * np.cos(r_beta) * np.cos(r_gamma)
    ) ** 0.5
    a_x, a_y, a_z = a_, b_ * np.cos(r_gamma), c_ * np.cos(r_beta)
    b_y = b_ * np.sin(r_gamma)
    b_z = c_ * (np.cos(r_alpha) - np.cos(r_beta) * np.cos(r_gamma)) / np.sin(r_gamma)
    c_z = volume / (a_ * b_ * np.sin(r_gamma))
    return np.array([[a_x, a_y, a_z], [0, b_y, b_z], [0, 0, c_z]])


def lattice_array_to_unit_cell(lattice_array) -> np.ndarray:
    """Reference utilities.py:693-709."""
    cell_lengths = np.sqrt(np.sum(lattice_array**2, axis=0))
    gamma_r = np.arccos(lattice_array[0][1] / cell_lengths[1])
    beta_r = np.arccos(lattice_array[0][2] / cell_lengths[2])
    alpha_r = np.arccos(lattice_array[1][2] * np.sin(gamma_r) / cell_lengths[2] + np.cos(beta_r) * np.cos(gamma_r))
    return np.append(cell_lengths, [np.rad2deg(alpha_r), np.rad2deg(beta_r), np.rad2deg(gamma_r)])


def system_lattice(system: dict):
    """(lattice, periodic) with the reference's mode rules (utilities.py:843-851, 889-893)."""
    if "unit_cell" in system and np.asarray(system["unit_cell"]).shape == (6,):
        periodic = True
    elif "unit_cell" not in system and "lattice" in system and np.asarray(system["lattice"]).shape == (3, 3):
        periodic = True
    else:
        periodic = False
    if not periodic:
        return None, False
    lattice = system["lattice"] if "lattice" in system else unit_cell_to_lattice_array(system["unit_cell"])
    return np.ascontiguousarray(lattice, dtype=np.float64), True


class CellTopology:
    """Per-atom constants of a system (the same for every frame of a trajectory)."""

    def __init__(self, elements, tol: float = 0.4):
        try:
            ids = element_ids(elements)
        except KeyError:
            raise
        self.n = len(ids)
        upper = [str(e).upper() for e in elements]
        self.cov = np.ascontiguousarray(COVALENT[ids])
        self.mass = np.ascontiguousarray(MASS[ids])
        self.terminal = np.array([u in TERMINAL_SYMBOLS for u in upper], dtype=np.uint8)
        # utilities.py:949-953: twice the largest covalent radius present plus the tolerance
        self.tol = float(tol)
        self.max_dist = 2 * max(atomic_covalent_radius[u] for u in set(upper)) + tol


def pack_frames(coords, lattices):
    """coords (F, N, 3); lattices (F, 3, 3) or None -> contiguous arrays + inverses."""
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    if lattices is None:
        return coords, None, None
    lat = np.ascontiguousarray(lattices, dtype=np.float64).reshape(-1, 3, 3)
    # fractional_from_cartesian inverts the lattice with numpy for every call (utilities.py:726); the
    # stacked call runs the same LAPACK routine on every 3 x 3 matrix
    inv = np.ascontiguousarray(np.linalg.inv(lat))
    return coords, lat, inv


def molecules_from_output(system: dict, n_mol, mol_offset, src_atom, out_xyz) -> list[dict]:
    """Output arrays of one frame -> the reference's list of dicts (utilities.py:1057-1067)."""
    el = np.asarray(system["elements"])
    ids = np.asarray(system["atom_ids"]) if "atom_ids" in system else None
    out = []
    for m in range(int(n_mol)):
        lo, hi = int(mol_offset[m]), int(mol_offset[m + 1])
        src = src_atom[lo:hi]
        d = {"elements": np.array(el[src], dtype="str"), "coordinates": np.array(out_xyz[lo:hi])}
        if ids is not None:
            d["atom_ids"] = np.array(ids[src], dtype="str")
        out.append(d)
    return out


def discrete_molecules_frames(topology: CellTopology, coords, lattices, rebuild: bool, device=None,
                              atoms_cap: int | None = None):
    """Run the kernel on F frames sharing one topology.  Returns
    ``(n_mol (F,), mol_offset (F, mols_cap+1), src_atom (F, cap), src_image (F, cap), xyz (F, cap, 3))``."""
    from . import engine

    coords, lat, inv = pack_frames(coords, lattices)
    if rebuild and lat is None:
        raise KeyError("lattice")
    return engine.context(device).discrete_molecules(topology, coords, lat, inv, bool(rebuild), atoms_cap)


def discrete_molecules(system: dict, rebuild=None, tol: float = 0.4, device=None) -> list[dict]:
    """Reference utilities.py:820-1085.  ``rebuild``: ``None``/``False`` or anything truthy (the
    reference passes the supercell dict; the 3x3x3 supercell is built on the device here)."""
    if "elements" not in system:
        from .trajectory import _FunctionError

        raise _FunctionError(
            "The 'elements' key is missing in the 'system' dictionary attribute of the MolecularSystem object. "
            "Which means, you need to decipher the forcefield based atom keys first (see manual)."
        )
    do_rebuild = rebuild is not None and rebuild is not False
    if do_rebuild:
        system["atom_ids"]   # KeyError like create_supercell (utilities.py:798)
    lattice, periodic = system_lattice(system)
    if do_rebuild and not periodic:
        lattice = system["lattice"] if "lattice" in system else unit_cell_to_lattice_array(system["unit_cell"])
    topo = CellTopology(system["elements"], tol)
    xyz = np.asarray(system["coordinates"], dtype=np.float64)[None]
    lat = None if lattice is None else np.asarray(lattice, dtype=np.float64)[None]
    n_mol, off, src, img, oxyz = discrete_molecules_frames(topo, xyz, lat, do_rebuild, device)
    return molecules_from_output(system, n_mol[0], off[0], src[0], oxyz[0])
