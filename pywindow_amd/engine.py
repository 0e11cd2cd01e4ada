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
_default_device: int | None = None


def set_default_device(device: int | None) -> None:
    """Pin the device used when none is given (``None``: back to automatic selection)."""
    global _default_device
    _default_device = None if device is None else int(device)


def resolve_device(device: int | None = None) -> int:
    """The HIP ordinal an analysis runs on when the caller names none.

    One process per GPU is the design point (reference: one worker per core,
    trajectory.py:553-586), so the choice follows what the process already said about its GPU,
    in this order: an explicit ``device`` argument; :func:`set_default_device`; the device PyTorch
    has made current (``torch.cuda.set_device(local_rank)`` of a ``torchrun`` job -- only consulted
    when the application has imported torch and initialised its GPU runtime); ``LOCAL_RANK`` (set by
    ``torchrun``) modulo the number of visible devices; device 0.
    """
    if device is not None:
        return int(device)
    if _default_device is not None:
        return _default_device
    import os
    import sys

    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                return int(torch.cuda.current_device())
        except Exception:  # pragma: no cover - a broken torch install must not break the engine
            pass
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None and lr.isdigit():
        n = _lib.load().pw_device_count()
        return int(lr) % n if n > 0 else int(lr)
    return 0


def context(device: int | None = None) -> "_lib.Context":
    """Lazily created per-device context (streams + workspaces); ``device=None``: :func:`resolve_device`."""
    dev = resolve_device(device)
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


def analyse(molecules, stages: int = _lib.STAGE_ALL, device: int | None = None, params=None, extra=None) -> np.ndarray:
    """Run the selected stages for a list of molecules in ONE launch; returns the
    structured record array (``_lib.UNIT_OUT_DTYPE``).  ``params``: ``_lib.Params`` for
    non-default find_windows / find_average_diameter knobs.  ``extra``: a list that receives the
    windows beyond ``_lib.W_MAX`` (see :func:`windows_of`)."""
    return context(device).analyse(make_batch(molecules), stages, params, extra)


def extra_by_unit(extra) -> dict:
    """``{unit: (diameters, centres)}`` from the ``EXTRA_WINDOW_DTYPE`` arrays an analysis appended to its
    ``extra`` list (entries are ordered by unit, then by position)."""
    out: dict = {}
    for arr in extra or ():
        for u in np.unique(arr["unit"]):
            rows = arr[arr["unit"] == u]
            out[int(u)] = (rows["d"].copy(), rows["c"].copy())
    return out


def windows_of(rec, more=None):
    """(diameters, centres) arrays or None, as ``find_windows`` returns them (utilities.py:1526-1553).
    ``more``: this unit's entry of :func:`extra_by_unit` -- a record holds ``_lib.W_MAX`` windows, the
    reference any number.  No survivors -> ``None``; survivors but only noise -> two empty arrays, the
    centres of shape ``(0,)`` like the reference's ``np.array([])``."""
    n = int(rec["n_windows"])
    if n < 0:
        return None
    if n == 0:
        return np.array([]), np.array([])
    k = min(n, _lib.W_MAX)
    d, c = np.array(rec["win_d"][:k]), np.array(rec["win_c"][:k]).reshape(k, 3)
    if n > k:
        if more is None or len(more[0]) != n - k:
            raise _lib.PwHipError(f"{n} windows but only {k} in the record: pass the analysis' extra-window list")
        d, c = np.concatenate([d, more[0]]), np.concatenate([c, more[1]])
    return d, c


def raise_on_capacity(rec) -> None:
    """A result that the engine could not compute is an error, never a value: more sampling vectors than
    the workspace of the launch held (``pw_analysis_batch`` grows it and repeats, so this only shows on
    the resident path) is raised here."""
    st = int(rec["status"])
    if st & _lib.ST_PATH_TOO_LONG:
        # (the reference builds the path of every sampling vector as a Python list, utilities.py:1100-1129)
        raise MemoryError(
            f"find_windows: a sampling sphere of radius {float(rec['sphere_r']):.6g} means more than 2**20 points per path "
            "scan -- the pore centre ran away (an open or enormous opt_pore_diameter box?)")
    if st & _lib.ST_POINTS_OVERFLOW:
        raise _lib.PwHipError(
            f"sampling-vector workspace too small (find_windows wants {int(rec['n_points'])}, find_average_diameter "
            f"{int(rec['n_points_avg'])}): analyse through pw_analysis_batch / Context.analyse, which grows it")


def raise_on_uncomputable(recs) -> None:
    """The resident paths (trajectory drivers) after their capacities have settled (``Resident.download_settled``
    grows the sampling-vector workspace and repeats the launch): a unit that STILL carries
    ``PW_ST_POINTS_OVERFLOW`` has no average diameter and no windows -- an error, never a value -- and a unit
    with fewer than ten sampling vectors makes the reference's trajectory analysis raise ``ValueError`` from
    ``KDTree.query(k=10)`` (utilities.py:1428-1431), so it does here."""
    st = recs["status"]
    bad = np.flatnonzero(st & (_lib.ST_POINTS_OVERFLOW | _lib.ST_PATH_TOO_LONG))
    if len(bad):
        raise_on_capacity(recs[bad[0]])
    if (st & _lib.ST_TOO_FEW_POINTS).any():
        raise ValueError("k must be less than or equal to the number of training points")


#: what scipy.optimize.minimize raises inside the reference's opt_pore_diameter / find_windows when the
#: pore radius at the start is negative: the box ``start -/+ r`` is inverted (utilities.py:412-424)
NEGATIVE_PORE_MESSAGE = "An upper bound is less than the corresponding lower bound."


def raise_like_reference(rec) -> None:
    """Single-molecule calls fail where the reference fails: a non-porous molecule makes
    ``opt_pore_diameter`` / ``find_windows(pore_opt=True)`` / ``full_analysis`` raise ``ValueError``
    from inside SciPy (bounds ``com -/+ r`` with ``r < 0``)."""
    if int(rec["status"]) & _lib.ST_NEGATIVE_PORE:
        raise ValueError(NEGATIVE_PORE_MESSAGE)


def warn_like_reference(rec) -> None:
    """The reference only logs these conditions (utilities.py:1538-1551).  Batched drivers also log
    a non-porous unit here (the reference would have stopped the whole trajectory with a ValueError;
    the unit carries ``PW_ST_NEGATIVE_PORE``, no optimised pore and ``None`` windows)."""
    st = int(rec["status"])
    if st & _lib.ST_NEGATIVE_PORE:
        logger.warning("pywindow_amd: pore radius at the centre of mass is not positive; the reference raises "
                       "ValueError('%s') here -- pore_diameter_opt and windows of this molecule are not computed.",
                       NEGATIVE_PORE_MESSAGE)
    if st & _lib.ST_WINDOW_DROPPED:
        logger.warning("Warning. One of the analysed windows has returned as None. See manual.")
    if st & _lib.ST_WINDOW_NEGATIVE:
        logger.warning(
            "Warning. One of the analysed windows has a vdW corrected diameter smaller than 0. See manual."
        )
    if st & _lib.ST_POINTS_OVERFLOW:
        logger.warning("pywindow_amd: sampling-vector workspace too small for this unit (status=%d): its average "
                       "diameter is NaN and its windows None -- NOT results.", st)
    if st & _lib.ST_TOO_FEW_POINTS:
        logger.warning("pywindow_amd: fewer than ten sampling vectors; the reference raises ValueError here "
                       "(KDTree.query(k=10)) -- the windows of this molecule are not computed.")


def offset_extra(extra: np.ndarray, offset: int) -> np.ndarray:
    """The extra-window entries of a piece of a batch, renumbered to the whole batch."""
    out = extra.copy()
    out["unit"] += offset
    return out


def records_to_properties(recs: np.ndarray, stages: int = _lib.STAGE_ALL, extra=None) -> list:
    """Many records -> the nested dicts of ``record_to_properties``, column-wise: every field is
    converted to Python objects once for the whole array (the per-record route spends its time in
    numpy scalar look-ups), and the conditions the reference logs are reported on the way.
    ``extra``: the ``EXTRA_WINDOW_DTYPE`` entries of these records (windows beyond ``_lib.W_MAX``)."""
    n = len(recs)
    if n == 0:
        return []
    more = extra_by_unit([extra]) if extra is not None and len(extra) else {}
    for rec in recs[recs["status"] != 0]:
        warn_like_reference(rec)
    n_atoms = recs["n_atoms"].tolist()
    com = recs["com"].copy()
    maxd, maxd_i, maxd_j = recs["maxd"].tolist(), recs["maxd_i"].tolist(), recs["maxd_j"].tolist()
    avg = recs["avg_d"].tolist()
    pore_d, pore_atom, pore_vol = recs["pore_d"].tolist(), recs["pore_atom"].tolist(), recs["pore_vol"].tolist()
    opt_d, opt_atom = recs["pore_opt_d"].tolist(), recs["pore_opt_atom"].tolist()
    opt_c, opt_vol = recs["pore_opt_c"].copy(), recs["pore_vol_opt"].tolist()
    n_win = recs["n_windows"].tolist()
    win_d, win_c = recs["win_d"], recs["win_c"]
    with_avg = bool(stages & _lib.STAGE_AVG)
    with_opt = bool(stages & (_lib.STAGE_OPT | _lib.STAGE_WINDOWS))
    with_win = bool(stages & _lib.STAGE_WINDOWS)
    out = []
    for i in range(n):
        props = {"no_of_atoms": n_atoms[i], "centre_of_mass": com[i],
                 "maximum_diameter": {"diameter": maxd[i], "atom_1": maxd_i[i], "atom_2": maxd_j[i]}}
        if with_avg:
            props["average_diameter"] = avg[i]
        props["pore_diameter"] = {"diameter": pore_d[i], "atom": pore_atom[i]}
        props["pore_volume"] = pore_vol[i]
        if with_opt:
            props["pore_diameter_opt"] = {"diameter": opt_d[i], "atom_1": opt_atom[i], "centre_of_mass": opt_c[i]}
            props["pore_volume_opt"] = opt_vol[i]
        if with_win:
            k = n_win[i]
            if k < 0:
                props["windows"] = {"diameters": None, "centre_of_mass": None}
            elif 0 < k <= _lib.W_MAX:
                props["windows"] = {"diameters": win_d[i, :k].copy(), "centre_of_mass": win_c[i, :k].copy()}
            else:           # none (two empty arrays, as the reference builds them) or more than a record holds
                d, c = windows_of(recs[i], more.get(i))
                props["windows"] = {"diameters": d, "centre_of_mass": c}
        out.append(props)
    return out


def record_to_properties(rec, stages: int = _lib.STAGE_ALL, more=None) -> dict:
    """One result record -> the nested dict ``Molecule.full_analysis()`` returns.  ``more``: the unit's
    windows beyond ``_lib.W_MAX`` as ``(diameters, centres)`` (:func:`extra_by_unit`)."""
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
        win = windows_of(rec, more)
        if win is None:
            props["windows"] = {"diameters": None, "centre_of_mass": None}
        else:
            props["windows"] = {"diameters": win[0], "centre_of_mass": win[1]}
    return props
