#!/usr/bin/env python
"""Where does the distance primitive's pinned summation order hold?  (diagnostic, CPU only)

The library restates sklearn's euclidean_distances with ONE order of the three-term dot product
for the N x N call, g = fma(z,z', fma(y,y', x*x')).  That order is what the BLAS under sklearn does for the bulk of
the matrix; this script finds the entries of a random N x N call where it does not, grouped by where the row / the
column sits in the matrix, and reports which of a family of candidate orders explains each group.

What it shows on this image (OpenBLAS 0.3.29, SkylakeX kernels): the order holds everywhere when N % 8 < 4 and for
small matrices; when N % 8 >= 4, some entries whose row or column is one of the four atoms [8*(N//8), 8*(N//8)+4)
come out one ulp different -- the edge kernel of those four sums its three products differently by column chunk.
The first mode of this script is how that was found (no single order of the family below explains the edge rows);
--rule checks the rule that does (edge_order below) entry by entry.  See DESIGN.md section 7.

    python tests/tools/distance_order_probe.py [sizes...]
    python tests/tools/distance_order_probe.py --rule [sizes...]     # the rule found, entry by entry (see edge_order)
"""
import itertools
import sys
from fractions import Fraction

import numpy as np
from sklearn.metrics.pairwise import euclidean_distances


def fma(a, b, c):
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def candidates(a, b):
    out = {}
    for i0, i1, i2 in itertools.permutations(range(3)):
        out["fma%d(fma%d(mul%d))" % (i2, i1, i0)] = fma(a[i2], b[i2], fma(a[i1], b[i1], a[i0] * b[i0]))
        out["fma%d(mul%d)+mul%d" % (i1, i0, i2)] = fma(a[i1], b[i1], a[i0] * b[i0]) + a[i2] * b[i2]
        if i0 < i1:
            out["(mul%d+mul%d)+mul%d" % (i0, i1, i2)] = (a[i0] * b[i0] + a[i1] * b[i1]) + a[i2] * b[i2]
    return out


def distance(g, xi, xj):
    return np.sqrt(max(fma(-2.0, g, xi) + xj, 0.0))


def last_panel_start(n):
    """Panels of 192 rows while at least 384 are left, then what is left in two halves rounded to 32, then the rest:
    the level-3 driver's recurrence with ONE BLAS thread (run this script with OPENBLAS_NUM_THREADS=1 beyond 382 atoms:
    from 383 OpenBLAS threads the product and the entries follow the thread count)."""
    start = 0
    while True:
        rem = n - start
        mi = 192 if rem >= 384 else (32 * ((rem // 2 + 31) // 32) if rem > 192 else rem)
        if start + mi >= n:
            return start
        start += mi


def edge_order(n, i, j):
    """The rule the library and the oracle restate (pw_unit.hpp: GramEdgeRule, pw_prim.c: edge_order)."""
    if n % 8 < 4:
        return False
    t0 = 8 * (n // 8)
    ei, ej = t0 <= i < t0 + 4, t0 <= j < t0 + 4
    if not (ei or ej):
        return False
    c = j if ei else i
    panel = last_panel_start(n)                               # where the last row panel starts (GEMM_P = 192)
    if c < panel:                                             # one kernel call for everything left of the panel
        return c < 12 * (panel // 12)
    w = min(32, n - 32 * (c // 32))
    return (c % 32) < 12 * (w // 12)


def check_rule(sizes):
    """Every entry of the edge rows, the rows after them and three body rows of numpy's X @ X.T against the rule."""
    rng = np.random.default_rng(7)
    for n in sizes:
        t0 = 8 * (n // 8)
        rows = list(range(t0, n)) + [0, 5 % n, max(t0 - 1, 0)]
        bad = tot = 0
        for rep in range(3):
            X = rng.normal(size=(n, 3)) * 5
            G = X @ X.T
            for i in rows:
                for j in range(n):
                    a, b = X[i], X[j]
                    if edge_order(n, i, j):
                        g = fma(a[2], b[2], a[0] * b[0] + a[1] * b[1])
                    else:
                        g = fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))
                    tot += 2
                    bad += (g != G[i, j]) + (g != G[j, i])
        print("N %4d (N %% 8 = %d): %6d entries, %d off the rule" % (n, n % 8, tot, bad))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--rule":
        check_rule([int(s) for s in sys.argv[2:]] or [12, 13, 20, 21, 23, 28, 29, 30, 31, 36, 44, 45, 47, 52, 60, 61, 63, 78, 92,
                                                      95, 100, 119, 124, 127, 140, 156, 159, 168, 170, 172, 175, 180, 183, 188, 191, 196, 204, 223, 255,
                                                      260, 287, 300, 316, 349, 380, 381, 382, 383, 412])
        return
    sizes = [int(s) for s in sys.argv[1:]] or [16, 20, 21, 45, 78, 100, 119, 168, 170]
    rng = np.random.default_rng(2)
    for n in sizes:
        X = np.round(rng.normal(size=(n, 3)) * 5, 6)
        D = euclidean_distances(X, X)
        XX = np.array([(x * x + z * z) + y * y for x, y, z in X])
        t0 = 8 * (n // 8)
        tail = range(t0, min(t0 + 4, n)) if n % 8 >= 4 else range(0)
        regions = {
            "body x body": [(i, j) for i in range(t0) for j in range(t0) if i != j],
            "edge-4 rows x body": [(i, j) for i in tail for j in range(t0)],
            "body x edge-4 columns": [(i, j) for i in range(t0) for j in tail],
            "edge-4 x edge-4": [(i, j) for i in tail for j in tail if i != j],
            "rows after the edge-4": [(i, j) for i in range(t0 + len(tail), n) for j in range(n) if i != j],
            "columns after the edge-4": [(i, j) for i in range(n) for j in range(t0 + len(tail), n) if i != j],
        }
        print("N = %d (N %% 8 = %d)" % (n, n % 8))
        for name, entries in regions.items():
            if not entries:
                continue
            fails = {}
            for i, j in entries:
                for k, g in candidates(X[i], X[j]).items():
                    fails[k] = fails.get(k, 0) + (distance(g, XX[i], XX[j]) != D[i, j])
            pinned = fails["fma2(fma1(mul0))"]
            best = sorted(fails.items(), key=lambda kv: kv[1])[:2]
            print("   %-26s %6d entries, the pinned order differs at %3d; best candidates %s" % (name, len(entries), pinned, best))


if __name__ == "__main__":
    main()
