// pw_shape.hpp -- shape descriptors of a molecule and the circumcircle estimate of a window
// (reference utilities.py:434-650 and :1653-1691), written against the team
// abstraction of pw_team.hpp like the unit pipeline.
//
// The tensors are sums, and the reference takes them with numpy: a sum over a contiguous array
// is numpy's pairwise reduction (blocks of <= 128 with eight accumulators, 8192-element buffers
// chained sequentially), a sum down the rows of an (N, 3) array is sequential.  Both are
// reproduced term by term, so the tensors are bit-identical to the reference's.  The inertia
// tensor's (N, 1) mass column against (N,) position rows broadcasts to N x N in the reference
// (utilities.py:511-522): every entry here is the same N^2-term sum, generated on the fly.
// Eigenvalues: the reference calls LAPACK's general dgeev; here a cyclic Jacobi iteration on
// the symmetric 3x3 (agreement to a few ulps of the largest eigenvalue, not bit-identical).
#pragma once
#include "pw_unit.hpp"

namespace pw {

constexpr int SHAPE_ROUND = 4;        // 8192-element buffers summed side by side
constexpr int SHAPE_LEAVES = 160;     // leaves of one buffer's pairwise recursion (<= 137)

struct ShapeScratch {                 // team-shared
    int loff[2][SHAPE_LEAVES], llen[2][SHAPE_LEAVES], nleaf[2];
    double leaf[SHAPE_ROUND][SHAPE_LEAVES];
    double part[SHAPE_ROUND];
    double total;
    double com[3], mw;
};

// numpy's leaf: eight strided accumulators, folded pairwise, then the tail
template <class F>
PW_HD inline double shape_leaf_sum(F f, long e0, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r = r + f(e0 + i);
        return r;
    }
    double r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = f(e0 + c);
    int lim = n - (n % 8);
    for (int i = 8; i < lim; i += 8) {
#pragma unroll
        for (int c = 0; c < 8; ++c) r[c] += f(e0 + i + c);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int i = lim; i < n; ++i) res = res + f(e0 + i);
    return res;
}

// np.sum over `count` contiguous float64 terms f(0..count-1); result valid on every thread
template <class T, class F>
PW_HD inline double shape_np_sum(ShapeScratch& sc, long count, F f) {
    const long nbuf = (count + 8191) / 8192;
    const int tail = (int)(count - (nbuf - 1) * 8192);
    if (T::tid() == 0) {
        sc.nleaf[0] = np_leaves(0, 8192, sc.loff[0], sc.llen[0]);
        sc.nleaf[1] = np_leaves(0, tail, sc.loff[1], sc.llen[1]);
        sc.total = 0.0;
    }
    T::sync();
    for (long b0 = 0; b0 < nbuf; b0 += SHAPE_ROUND) {
        const int nb = (int)(nbuf - b0 < SHAPE_ROUND ? nbuf - b0 : SHAPE_ROUND);
        for (int task = T::tid(); task < nb * SHAPE_LEAVES; task += T::SIZE) {
            int b = task / SHAPE_LEAVES, l = task % SHAPE_LEAVES;
            int which = (b0 + b == nbuf - 1) ? 1 : 0;
            if (l < sc.nleaf[which])
                sc.leaf[b][l] = shape_leaf_sum(f, (b0 + b) * 8192 + sc.loff[which][l], sc.llen[which][l]);
        }
        T::sync();
        for (int b = T::tid(); b < nb; b += T::SIZE) {
            int which = (b0 + b == nbuf - 1) ? 1 : 0;
            sc.part[b] = np_combine(which ? tail : 8192, sc.leaf[b]);
        }
        T::sync();
        if (T::tid() == 0) {
            double t = sc.total;
            for (int b = 0; b < nb; ++b) t = (b0 + b == 0) ? sc.part[b] : t + sc.part[b];
            sc.total = t;
        }
        T::sync();
    }
    double r = sc.total;
    T::sync();
    return r;
}

// eigenvalues of a symmetric 3x3, descending (cyclic Jacobi)
PW_HD inline void shape_eigenvalues(const double m[3][3], double* ev) {
    double a[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) a[i][j] = m[i][j];
    for (int sweep = 0; sweep < 60; ++sweep) {
        double offd = pw_abs(a[0][1]) + pw_abs(a[0][2]) + pw_abs(a[1][2]);
        double diag = pw_abs(a[0][0]) + pw_abs(a[1][1]) + pw_abs(a[2][2]);
        if (offd == 0.0 || offd <= 1e-300 + 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (pw_abs(theta) + pw_sqrt(theta * theta + 1.0));
                double c = 1.0 / pw_sqrt(t * t + 1.0), s = t * c;
                a[p][p] -= t * apq;
                a[q][q] += t * apq;
                a[p][q] = a[q][p] = 0.0;
                int r = 3 - p - q;
                double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
            }
    }
    double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2], t;
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 < e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    ev[0] = e0; ev[1] = e1; ev[2] = e2;
}

// one molecule: xyz (n, 3) row-major, per-atom masses
template <class T>
PW_HD inline void shape_unit(ShapeScratch& sc, const double* xyz, const double* mass, int n, pw_shape_out* out) {
    const long N = n;
    // ---- centre of mass (utilities.py:127-148) ----
    if (T::tid() == 0) {
        double tot = 0.0;
        for (int s0 = 0; s0 < n; s0 += 8192) {
            double part = np_sum_small(mass + s0, n - s0 < 8192 ? n - s0 : 8192);
            tot = s0 == 0 ? part : tot + part;
        }
        sc.mw = tot;
    }
    T::sync();
    for (int c = T::tid(); c < 3; c += T::SIZE)
        sc.com[c] = seq_sum_blocked(n, [&](int i) { return xyz[3 * i + c] * mass[i]; }) / sc.mw;
    T::sync();
    const double cx = sc.com[0], cy = sc.com[1], cz = sc.com[2];
    // ---- gyration tensor (utilities.py:461-495) ----
    double g[6];
    for (int c = T::tid(); c < 3; c += T::SIZE) {
        double cc = sc.com[c];
        sc.part[c] = seq_sum_blocked(n, [&](int i) { double d = xyz[3 * i + c] - cc; return d * d; });
    }
    T::sync();
    g[0] = sc.part[0]; g[1] = sc.part[1]; g[2] = sc.part[2];
    T::sync();
    g[3] = shape_np_sum<T>(sc, N, [&](long i) { return (xyz[3 * i] - cx) * (xyz[3 * i + 1] - cy); });
    g[4] = shape_np_sum<T>(sc, N, [&](long i) { return (xyz[3 * i] - cx) * (xyz[3 * i + 2] - cz); });
    g[5] = shape_np_sum<T>(sc, N, [&](long i) { return (xyz[3 * i + 1] - cy) * (xyz[3 * i + 2] - cz); });
    // ---- inertia tensor (utilities.py:498-529): N x N terms mass_i * f(position_j) ----
    auto sq = [&](long j, int c) { double v = xyz[3 * j + c]; return v * v; };
    double t[6];
    t[0] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return mass[i] * (sq(j, 1) + sq(j, 2)); });
    t[1] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return mass[i] * (sq(j, 0) + sq(j, 2)); });
    t[2] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return mass[i] * (sq(j, 0) + sq(j, 1)); });
    t[3] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return ((-mass[i]) * xyz[3 * j]) * xyz[3 * j + 1]; });
    t[4] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return ((-mass[i]) * xyz[3 * j]) * xyz[3 * j + 2]; });
    t[5] = shape_np_sum<T>(sc, N * N, [&](long e) { long i = e / N, j = e - i * N; return ((-mass[i]) * xyz[3 * j + 1]) * xyz[3 * j + 2]; });
    if (T::tid() == 0) {
        const double dn = (double)n;
        double G[3][3] = {{g[0], g[3], g[4]}, {g[3], g[1], g[5]}, {g[4], g[5], g[2]}};
        double I[3][3] = {{t[0], t[3], t[4]}, {t[3], t[1], t[5]}, {t[4], t[5], t[2]}};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                G[i][j] = G[i][j] / dn;
                I[i][j] = I[i][j] / dn;
                out->gyration[i][j] = G[i][j];
                out->inertia[i][j] = I[i][j];
            }
        double ev[3];
        shape_eigenvalues(I, ev);
        out->eigenvalues[0] = ev[0]; out->eigenvalues[1] = ev[1]; out->eigenvalues[2] = ev[2];
        // utilities.py:434-446
        out->asphericity = ev[0] - (ev[1] + ev[2]) / 2.0;
        out->acylidricity = ev[1] - ev[2];
        double tr = (ev[0] + ev[1]) + ev[2];
        out->relative_shape_anisotropy =
            1.0 - 3.0 * ((((ev[0] * ev[1]) + (ev[0] * ev[2])) + (ev[1] * ev[2])) / pw_square_np(tr));
    }
    T::sync();
}

// utilities.py:1653-1676 -- one atom triple of one molecule
PW_HD inline void circumcircle_one(const double* xyz, const int* set, double* diameter, double* centre) {
    const double* A = xyz + 3 * (long)set[0];
    const double* B = xyz + 3 * (long)set[1];
    const double* C = xyz + 3 * (long)set[2];
    double a = norm3(C[0] - B[0], C[1] - B[1], C[2] - B[2]);
    double b = norm3(C[0] - A[0], C[1] - A[1], C[2] - A[2]);
    double c = norm3(B[0] - A[0], B[1] - A[1], B[2] - A[2]);
    double s = ((a + b) + c) / 2.0;
    double r = (((a * b) * c) / 4.0) / pw_sqrt(((s * (s - a)) * (s - b)) * (s - c)) - 1.70;
    double b1 = (a * a) * (((b * b) + (c * c)) - (a * a));
    double b2 = (b * b) * (((a * a) + (c * c)) - (b * b));
    double b3 = (c * c) * (((a * a) + (b * b)) - (c * c));
    double den = (b1 + b2) + b3;
    for (int k = 0; k < 3; ++k) {
        // row k of column_stack((A, B, C)) against (b1, b2, b3), BLAS dgemv order
        double v = pw_fma(C[k], b3, pw_fma(A[k], b1, B[k] * b2));
        centre[k] = v / den;
    }
    *diameter = r * 2.0;
}

}  // namespace pw
