"""The scans that skip what cannot matter (round 6: a wave's path scans through the atoms that can undercut the gap at the
origin -- wave_path_candidates --, DBSCAN's adjacency rows without the blocks out of reach in z) against the same source
compiled WITHOUT them (-DPW_NO_SCAN_LISTS): every field of every record must be the same bits, on molecules made to
stress the bounds -- shells, blobs, two shells, lattice fragments with many equal distances, rings, far from the origin."""
import ctypes

import numpy as np

from pywindow_amd import _lib, element_data as E, synth


def _molecules(n_mol, seed, max_atoms=220):
    rng = np.random.default_rng(seed)
    pool = np.array(["C", "H", "N", "O", "S", "F", "Cl", "Br"])
    out = []
    for _ in range(n_mol):
        n = int(rng.integers(4, max_atoms + 1))
        kind = int(rng.integers(0, 5))
        p = rng.normal(size=(n, 3))
        if kind == 0:
            p = p / np.linalg.norm(p, axis=1)[:, None] * rng.uniform(3.0, 12.0) + rng.normal(scale=rng.uniform(0.0, 0.5), size=(n, 3))
        elif kind == 1:
            p = p * rng.uniform(0.5, 6.0)
        elif kind == 2:
            r = np.where(rng.random(n) < 0.5, rng.uniform(4.0, 7.0), rng.uniform(9.0, 12.0))
            p = p / np.linalg.norm(p, axis=1)[:, None] * r[:, None]
            p[: n // 2] += rng.normal(scale=1.5, size=3)
        elif kind == 3:
            g = np.array([(x, y, z) for x in range(-3, 4) for y in range(-3, 4) for z in range(-3, 4)], dtype=np.float64) * 1.6
            g = g[np.linalg.norm(g, axis=1) > 3.0]
            p = g[rng.permutation(len(g))[: min(n, len(g))]]
        else:
            t = rng.uniform(0, 2 * np.pi, n)
            p = np.stack([np.cos(t) * 8.0, np.sin(t) * 8.0, rng.normal(scale=1.0, size=n)], axis=1) + rng.normal(scale=0.4, size=(n, 3))
        el = pool[rng.integers(0, int(rng.integers(1, len(pool) + 1)), size=len(p))]
        out.append((el, np.ascontiguousarray(p + rng.normal(scale=rng.choice([0.0, 5.0, 500.0]), size=3))))
    return out


def _run(lib, mols):
    off = np.zeros(len(mols) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(el) for el, _ in mols])
    ids = [E.element_ids(el) for el, _ in mols]
    xyz = np.ascontiguousarray(np.concatenate([p for _, p in mols]))
    vdw = np.ascontiguousarray(np.concatenate([E.VDW[i] for i in ids]))
    mass = np.ascontiguousarray(np.concatenate([E.MASS[i] for i in ids]))
    out = np.zeros(len(mols), dtype=_lib.UNIT_OUT_DTYPE)
    vp = ctypes.c_void_p
    rc = lib.hs_analysis_batch(ctypes.c_long(len(mols)), off.ctypes.data_as(vp), xyz.ctypes.data_as(vp), vdw.ctypes.data_as(vp),
                               mass.ctypes.data_as(vp), ctypes.c_uint(15), out.ctypes.data_as(vp), None)
    assert rc == 0
    return out


def test_records_with_and_without_the_scan_lists_are_the_same_bits(hostsim):
    with_lists = ctypes.CDLL(str(hostsim / "libunitprobe.so"))
    without = ctypes.CDLL(str(hostsim / "libunitprobe_nolists.so"))
    elements, frames = synth.synthetic_units(6)
    mols = [(elements, np.ascontiguousarray(f)) for f in frames] + _molecules(400, 2026)
    a, b = _run(with_lists, mols), _run(without, mols)
    assert (a["n_windows"] > 0).sum() >= 60 and (a["n_survivors"] > 0).sum() >= 120      # (the scans had work to do)
    assert a.tobytes() == b.tobytes()
