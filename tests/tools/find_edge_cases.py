#!/usr/bin/env python3
"""Search for inputs on which find_windows drops a window (refined path scan fails, utilities.py:1226-1227)
or returns a negative window diameter (:1545-1551), using the kernel source compiled for the host
(tests/hostsim, milliseconds per molecule).  The hits are written to tests/golden/cliff_extra_cases.npz;
tests/golden/make_golden.py cliffs then runs the REFERENCE on them and stores what it returns and logs.
Development tool: needs nothing but the repository."""
import ctypes
import json
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from _util import GOLDEN, load_group, molecules  # noqa: E402
from pywindow_amd import _lib  # noqa: E402
from pywindow_amd import element_data as E  # noqa: E402

L = ctypes.CDLL(str(ROOT / "tests" / "hostsim" / "libunitprobe.so"))


def run(el, xyz, prm):
    ids = E.element_ids(el)
    vdw, mass = np.ascontiguousarray(E.VDW[ids]), np.ascontiguousarray(E.MASS[ids])
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    off = np.array([0, len(xyz)], np.int64)
    out = np.zeros(1, dtype=_lib.UNIT_OUT_DTYPE)
    vp = ctypes.c_void_p
    rc = L.hs_analysis_batch(ctypes.c_long(1), off.ctypes.data_as(vp), xyz.ctypes.data_as(vp), vdw.ctypes.data_as(vp),
                             mass.ctypes.data_as(vp), ctypes.c_uint(_lib.STAGE_WINDOWS), out.ctypes.data_as(vp),
                             ctypes.byref(prm))
    assert rc == 0
    return out[0]


def main():
    rng = np.random.default_rng(7)
    pool = []
    for tag in ("static", "md20"):
        g = load_group(tag)
        for name, (el, xyz) in zip(g["names"], molecules(g)):
            pool.append((str(name), np.array(el), np.array(xyz, float)))
    found = {"dropped": [], "negative": []}
    # dropped windows: coarse path scans with a long step let vectors through that the 0.1 A scan rejects
    for name, el, xyz in pool:
        for inc in (2.5, 5.0, 8.0, 12.0):
            for noise in (0.0, 0.2, 0.4):
                for trial in range(3 if noise else 1):
                    c = np.round(xyz + rng.normal(0, noise, xyz.shape), 4) if noise else xyz
                    r = run(el, c, _lib.Params(increment=inc))
                    if int(r["status"]) & _lib.ST_WINDOW_DROPPED and len(found["dropped"]) < 3:
                        found["dropped"].append((f"{name}_inc{inc}_n{noise}_{trial}", el, c, {"increment": inc}))
        if len(found["dropped"]) >= 3:
            break
    # negative windows: the ray test of the pre-analysis starts at the centroid, the path scans at the pore
    # centre, so a path can graze an atom the ray misses; with coarse steps in both path scans (increment,
    # and window_analysis's increment2) such a vector survives and its neck -- found by the z search
    # between two samples -- is narrower than the atoms
    for name, el, xyz in pool:
        for inc, inc2 in ((5.0, 2.0), (8.0, 3.0), (5.0, 1.0), (12.0, 4.0)):
            for trial in range(12):
                c = np.round(xyz + rng.normal(0, 0.45, xyz.shape), 4)
                r = run(el, c, _lib.Params(increment=inc, increment2=inc2))
                st = int(r["status"])
                if st & _lib.ST_WINDOW_NEGATIVE and not st & _lib.ST_NEGATIVE_PORE and len(found["negative"]) < 3:
                    found["negative"].append((f"{name}_inc{inc}_inc2_{inc2}_{trial}", el, c, {"increment": inc, "increment2": inc2}))
        if len(found["negative"]) >= 3:
            break
    out = {"names": [], "labels": [], "mol": [], "kwargs": []}
    arrays = {}
    for kind, items in found.items():
        for name, el, c, kw in items:
            out["names"].append(name)
            out["labels"].append(f"{name}/win/{kind}")
            out["mol"].append(name)
            out["kwargs"].append(json.dumps(kw))
            arrays[f"{name}__el"] = np.array(el)
            arrays[f"{name}__xyz"] = c
            print(kind, name, len(el), kw)
    np.savez_compressed(GOLDEN / "cliff_extra_cases.npz", names=np.array(out["names"]), labels=np.array(out["labels"]),
                        mol=np.array(out["mol"]), kwargs=np.array(out["kwargs"]), **arrays)


if __name__ == "__main__":
    main()
