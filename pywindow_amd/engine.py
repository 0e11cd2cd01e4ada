"""Batched execution on the MI355X engine and conversion to the reference's
``Molecule.properties`` schema (molecular.py:215-352)."""

from __future__ import annotations

import logging
import threading

import numpy as np

from . import _lib
from .element_data import MASS, VDW, element_ids

logger = logging.getLogger("pywindow_amd")

_contexts: dict[int, "_lib.Context"] = {}
_lock = threading.Lock()
_default_device = 0


def set_default_device(device: int) -> None:
    global _default_device
    _default_device = int(device)


def context(device: int | None = None) -> "_lib.Context":
    """Lazily created per-device context (streams + workspaces)."""
    dev = _default_device if device is None else int(device)
    with _lock:
        ctx = _contexts.get(dev)
        if ctx is None:
            ctx = _lib.Context(dev)
            _contexts[dev] = ctx
        return ctx


def make_batch(molecules) -> "_lib.Batch":
    """``molecules``: iterable of (elements, coordinates (N,3)).  Element symbols
    are looked up exactly like the reference (upper-cased; KeyError if unknown)."""
    offs = [0]
    xyz = []
    vdw = []
    mass = []
    for elements, coords in molecules:
        ids = element_ids(elements)
        c = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        if len(c) != len(ids):
            raise ValueError("elements and coordinates differ in length")
        offs.append(offs[-1] + len(ids))
        xyz.append(c)
        vdw.append(VDW[ids])
        mass.append(MASS[ids])
    if not xyz:
        return _lib.Batch(np.zeros(1, np.int64), np.zeros((0, 3)), np.zeros(0), np.zeros(0))
    return _lib.Batch(np.array(offs, np.int64), np.concatenate(xyz), np.concatenate(vdw), np.concatenate(mass))


def analyse(molecules, stages: int = _lib.STAGE_ALL, device: int | None = None, params=None) -> np.ndarray:
    """Run the selected stages for a list of molecules in ONE launch; returns the
    structured record array (``_lib.UNIT_OUT_DTYPE``).  ``params``: ``_lib.Params`` for
    non-default find_windows / find_average_diameter knobs."""
    return context(device).analyse(make_batch(molecules), stages, params)


def windows_of(rec):
    """(diameters, centres) arrays or None, as ``find_windows`` returns them."""
    n = int(rec["n_windows"])
    if n < 0:
        return None
    return np.array(rec["win_d"][:n]), np.array(rec["win_c"][:n]).reshape(n, 3)


def warn_like_reference(rec) -> None:
    """The reference only logs these conditions (utilities.py:1538-1551)."""
    st = int(rec["status"])
    if st & _lib.ST_WINDOW_DROPPED:
        logger.warning("Warning. One of the analysed windows has returned as None. See manual.")
    if st & _lib.ST_WINDOW_NEGATIVE:
        logger.warning(
            "Warning. One of the analysed windows has a vdW corrected diameter smaller than 0. See manual."
        )
    if st & (_lib.ST_WINDOW_OVERFLOW | _lib.ST_POINTS_OVERFLOW):
        logger.warning("pywindow_amd: workspace limit hit (status=%d); results truncated.", st)


def record_to_properties(rec, stages: int = _lib.STAGE_ALL) -> dict:
    """One result record -> the nested dict ``Molecule.full_analysis()`` returns."""
    props: dict = {"no_of_atoms": int(rec["n_atoms"])}
    props["centre_of_mass"] = np.array(rec["com"])
    props["maximum_diameter"] = {
        "diameter": float(rec["maxd"]),
        "atom_1": int(rec["maxd_i"]),
        "atom_2": int(rec["maxd_j"]),
    }
    if stages & _lib.STAGE_AVG:
        props["average_diameter"] = float(rec["avg_d"])
    props["pore_diameter"] = {"diameter": float(rec["pore_d"]), "atom": int(rec["pore_atom"])}
    props["pore_volume"] = float(rec["pore_vol"])
    if stages & (_lib.STAGE_OPT | _lib.STAGE_WINDOWS):
        props["pore_diameter_opt"] = {
            "diameter": float(rec["pore_opt_d"]),
            "atom_1": int(rec["pore_opt_atom"]),
            "centre_of_mass": np.array(rec["pore_opt_c"]),
        }
        props["pore_volume_opt"] = float(rec["pore_vol_opt"])
    if stages & _lib.STAGE_WINDOWS:
        win = windows_of(rec)
        if win is None:
            props["windows"] = {"diameters": None, "centre_of_mass": None}
        else:
            props["windows"] = {"diameters": win[0], "centre_of_mass": win[1]}
    return props
