"""Synthetic DL_POLY trajectories for the benchmark configurations.

SURVEY.md section 8(d), config 2: frame ``k`` of the synthetic trajectory is the
CC3 cage of the reference's known-answer test (tests/test_validate_cc3.py:5-350,
kept here as the data file ``data/cc3_base.xyz``) plus iid N(0, sigma) noise per
coordinate drawn from ``numpy.random.default_rng(20260000 + k)``.  The frames
are serialised as DL_POLY HISTORY text (keytrj=0, imcon=0, ``%12.4E``
coordinates; layout of examples/data/input/HISTORY_singlemol_short) and every
consumer works from the *parsed* text, because the 5-significant-digit format
quantises the coordinates.
"""

from __future__ import annotations

import pathlib

import numpy as np

SEED_BASE = 20260000
_DATA = pathlib.Path(__file__).resolve().parent / "data"


def load_cc3_base() -> tuple[np.ndarray, np.ndarray]:
    """Return ``(elements (N,) str, coordinates (N,3) f64)`` of the CC3 cage."""
    elements = []
    xyz = []
    with (_DATA / "cc3_base.xyz").open() as fh:
        n = int(fh.readline())
        fh.readline()
        for _ in range(n):
            tok = fh.readline().split()
            elements.append(tok[0])
            # repr-precision text: float() round-trips the f64 exactly
            xyz.append((float(tok[1]), float(tok[2]), float(tok[3])))
    return np.array(elements), np.array(xyz, dtype=np.float64)


def noisy_frame(base: np.ndarray, seed: int, sigma: float = 0.10) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return base + rng.normal(0.0, sigma, size=base.shape)


def _fmt_e(v: float) -> str:
    return "%12.4E" % v


def history_text(
    elements,
    frames,
    title: str = "synthetic CC3 trajectory (pywindow_amd.synth)",
    tstep: float = 0.0007,
    cell=None,
) -> str:
    """Serialise frames (iterable of (N,3) arrays) as a keytrj=0 HISTORY; ``cell`` (3,3), rows =
    cell vectors, makes it a periodic (imcon=1 cubic / 3 otherwise) trajectory."""
    elements = list(elements)
    natms = len(elements)
    imcon = 0
    if cell is not None:
        cell = np.asarray(cell, dtype=float)
        cubic = np.allclose(cell, np.diag(np.diag(cell))) and np.allclose(np.diag(cell), cell[0, 0])
        imcon = 1 if cubic else 3
    out = [title, "%10d%10d%10d" % (0, imcon, natms)]
    for k, xyz in enumerate(frames):
        out.append(
            "timestep%10d%10d%10d%10d%12.6f" % (k + 1, natms, 0, imcon, tstep)
        )
        if cell is not None:
            for row in cell:
                out.append("%20.10f%20.10f%20.10f" % tuple(row))
        for i, el in enumerate(elements):
            out.append("%-8s%10d%12.6f%12.6f" % (el, i + 1, 0.0, 0.0))
            out.append(_fmt_e(xyz[i, 0]) + _fmt_e(xyz[i, 1]) + _fmt_e(xyz[i, 2]))
    return "\n".join(out) + "\n"


def write_history(path, elements, frames, title: str = "synthetic trajectory (pywindow_amd.synth)",
                  tstep: float = 0.0007, cell=None) -> pathlib.Path:
    """Stream frames (an iterable of (N,3) arrays) into a keytrj=0 HISTORY file, one frame's text at a
    time -- the same bytes as ``history_text``, without holding a long trajectory's text in memory."""
    elements = list(elements)
    natms = len(elements)
    imcon = 0
    cell_lines = ""
    if cell is not None:
        cell = np.asarray(cell, dtype=float)
        cubic = np.allclose(cell, np.diag(np.diag(cell))) and np.allclose(np.diag(cell), cell[0, 0])
        imcon = 1 if cubic else 3
        cell_lines = "".join("%20.10f%20.10f%20.10f\n" % tuple(row) for row in cell)
    keys = ["%-8s%10d%12.6f%12.6f\n" % (el, i + 1, 0.0, 0.0) for i, el in enumerate(elements)]
    path = pathlib.Path(path)
    with path.open("w") as fh:
        fh.write(title + "\n" + "%10d%10d%10d\n" % (0, imcon, natms))
        for k, xyz in enumerate(frames):
            rows = np.asarray(xyz, dtype=float).tolist()
            body = [None] * (2 * natms)
            body[0::2] = keys
            body[1::2] = ["%12.4E%12.4E%12.4E\n" % (r[0], r[1], r[2]) for r in rows]
            fh.write("timestep%10d%10d%10d%10d%12.6f\n" % (k + 1, natms, 0, imcon, tstep) + cell_lines + "".join(body))
    return path


def write_history_cycled(path, elements, distinct_frames, n_frames: int,
                         title: str = "synthetic trajectory, distinct frames cycled (pywindow_amd.synth)",
                         tstep: float = 0.0007, cell=None) -> pathlib.Path:
    """A LONG keytrj=0 HISTORY file in seconds: the text of ``distinct_frames`` (a list of (N,3) arrays) is
    formatted once and the file is ``n_frames`` "timestep" records that cycle through those bodies -- the same layout
    and the same parsing / analysis work per frame as ``write_history``, without formatting ten million lines in
    Python (a 10 000-frame, 1344-atom trajectory is 1 GB of text)."""
    elements = list(elements)
    natms = len(elements)
    imcon = 0
    cell_lines = ""
    if cell is not None:
        cell = np.asarray(cell, dtype=float)
        cubic = np.allclose(cell, np.diag(np.diag(cell))) and np.allclose(np.diag(cell), cell[0, 0])
        imcon = 1 if cubic else 3
        cell_lines = "".join("%20.10f%20.10f%20.10f\n" % tuple(row) for row in cell)
    keys = ["%-8s%10d%12.6f%12.6f\n" % (el, i + 1, 0.0, 0.0) for i, el in enumerate(elements)]
    bodies = []
    for xyz in distinct_frames:
        rows = np.asarray(xyz, dtype=float).tolist()
        body = [None] * (2 * natms)
        body[0::2] = keys
        body[1::2] = ["%12.4E%12.4E%12.4E\n" % (r[0], r[1], r[2]) for r in rows]
        bodies.append((cell_lines + "".join(body)).encode())
    path = pathlib.Path(path)
    with path.open("wb") as fh:
        fh.write((title + "\n" + "%10d%10d%10d\n" % (0, imcon, natms)).encode())
        for k in range(n_frames):
            fh.write(("timestep%10d%10d%10d%10d%12.6f\n" % (k + 1, natms, 0, imcon, tstep)).encode())
            fh.write(bodies[k % len(bodies)])
    return path


def write_synthetic_history(
    path,
    n_frames: int,
    sigma: float = 0.10,
    seed_base: int = SEED_BASE,
) -> pathlib.Path:
    """Write the config-2 style trajectory (CC3 + noise) to ``path``."""
    elements, base = load_cc3_base()
    frames = (noisy_frame(base, seed_base + k, sigma) for k in range(n_frames))
    path = pathlib.Path(path)
    path.write_text(history_text(elements, frames))
    return path


def quantise_like_history(xyz: np.ndarray) -> np.ndarray:
    """Coordinates after a ``%12.4E`` text round trip (what a parser would see)."""
    flat = np.array([float(_fmt_e(v)) for v in np.asarray(xyz).ravel()])
    return flat.reshape(np.asarray(xyz).shape)


def synthetic_units(
    n_frames: int,
    sigma: float = 0.10,
    seed_base: int = SEED_BASE,
    first: int = 0,
) -> tuple[np.ndarray, np.ndarray]:
    """In-memory equivalent of writing + parsing the synthetic HISTORY.

    Returns ``(elements (N,), coordinates (n_frames, N, 3))`` with the text
    quantisation applied, for frames ``first .. first+n_frames-1``.
    """
    elements, base = load_cc3_base()
    out = np.empty((n_frames,) + base.shape)
    for k in range(n_frames):
        out[k] = quantise_like_history(noisy_frame(base, seed_base + first + k, sigma))
    return elements, out


def screen_units(n_units: int, first: int = 0, frames_per_cage: int = 100, sigma: float = 0.10,
                 seed_base: int = SEED_BASE) -> tuple[np.ndarray, np.ndarray]:
    """BASELINE config 5 (combinatorial screen: random-perturbed cages x frames): unit u is frame u % frames_per_cage
    of cage u // frames_per_cage, CC3 + N(0, sigma) with seed ``seed_base + 1000 * cage + frame`` -- so
    a unit's coordinates do not depend on how the screen is sharded.  No text round trip (a screen's coordinates are
    not read from a HISTORY file).  Returns ``(elements, coordinates (n_units, N, 3))`` for units first ..."""
    elements, base = load_cc3_base()
    out = np.empty((n_units,) + base.shape)
    for i in range(n_units):
        u = first + i
        cage, frame = divmod(u, frames_per_cage)
        out[i] = base + np.random.default_rng(seed_base + 1000 * cage + frame).normal(0.0, sigma, size=base.shape)
    return elements, out


def threshold_cell():
    """Carbon pairs whose separations sit 3e-7 either side of the limits of the bond test (Rcov sum -+ tol = 0.96 /
    1.76, the latter also max_dist), along an axis, along a diagonal and through a cell face, beside pairs that are
    clearly bonded / clearly apart: the pairs whose test cannot be made once per frame."""
    L = 20.0
    lo, hi = 0.96, 1.76
    sites = []
    d111 = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    d120 = np.array([1.0, 2.0, 0.0]) / np.sqrt(5.0)
    k = 0
    for r in (hi - 3e-7, hi + 3e-7, lo + 3e-7, lo - 3e-7, hi - 5e-6, hi + 5e-6, 1.4, 0.5 * (lo + hi) + 1e-9,
              hi - 1e-8, hi + 1e-8, lo + 1e-8, lo - 1e-8):
        for u in (np.array([1.0, 0.0, 0.0]), d111, d120):
            base = np.array([2.5 + 5.0 * (k % 3), 2.5 + 5.0 * ((k // 3) % 3), 2.5 + 4.0 * (k // 9)])
            sites.append(base)
            sites.append(base + r * u)
            k += 1
    # through the faces: the partner is an image
    for m, r in enumerate((hi - 3e-7, hi + 3e-7, 1.5)):
        sites.append(np.array([0.3, 17.5, 18.5 - 4.0 * m]))
        sites.append(np.array([L - (r - 0.3), 17.5, 18.5 - 4.0 * m]))
    xyz = np.round(np.array(sites), 8)
    n = len(xyz)
    return {"elements": np.array(["C"] * n), "atom_ids": np.array(["C"] * n), "coordinates": xyz,
            "unit_cell": np.array([L, L, L, 90.0, 90.0, 90.0]), "lattice": np.eye(3) * L}
