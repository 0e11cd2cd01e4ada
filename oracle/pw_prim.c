/*
 * oracle/pw_prim.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Scalar C restatement of the one arithmetic primitive everything on the
 * pywindow hot path is built from: sklearn.metrics.pairwise.euclidean_distances
 * as the reference calls it at utilities.py:366 (N x N, "X is Y"), :384 and
 * :1116 (N x 1).  sklearn 1.6/1.7 (sklearn/metrics/pairwise.py:391-440)
 * computes   d = sqrt(max((-2 * X.Y^T + |x|^2) + |y|^2, 0))
 * with |x|^2 from einsum("ij,ij->i") and X.Y^T from OpenBLAS dgemm/dgemv; the
 * exact association below was established bit-for-bit against that stack
 * (OpenBLAS 0.3.29 AVX-512 kernels) -- SURVEY.md section 8a-0 -- and is
 * re-checked by tests/test_oracle.py against captured objective values in the
 * golden fixtures.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library (through oracle/pw_oracle.py).
 */
#include <math.h>
#include <stdint.h>

/* |x|^2 the way numpy's einsum("ij,ij->i") sums three products */
static inline double sqnorm3(const double *x) {
    return (x[0] * x[0] + x[2] * x[2]) + x[1] * x[1];
}

void pwo_row_sqnorms(int64_t n, const double *xyz, double *xx) {
    for (int64_t i = 0; i < n; ++i) xx[i] = sqnorm3(xyz + 3 * i);
}

/* distance atom i -> point p (N x 1 call shape: dgemv association) */
static inline double dist_point(const double *x, double xx, const double *p, double pp) {
    double g = fma(x[2], p[2], fma(x[0], p[0], x[1] * p[1]));
    double d2 = ((-2.0 * g) + xx) + pp;
    return sqrt(d2 > 0.0 ? d2 : 0.0);
}

/* min_i (|r_i - p| - vdw_i), first index on ties: pore_diameter()/2 (utilities.py:375-388) */
double pwo_min_gap(int64_t n, const double *xyz, const double *xx, const double *vdw,
                   const double *p, int64_t *argmin) {
    double pp = sqnorm3(p);
    double best = INFINITY;
    int64_t bi = 0;
    for (int64_t i = 0; i < n; ++i) {
        double v = dist_point(xyz + 3 * i, xx[i], p, pp) - vdw[i];
        if (v < best) { best = v; bi = i; }
    }
    if (argmin) *argmin = bi;
    return best;
}

/* all N gaps, for tests */
void pwo_gaps(int64_t n, const double *xyz, const double *xx, const double *vdw,
              const double *p, double *out) {
    double pp = sqnorm3(p);
    for (int64_t i = 0; i < n; ++i)
        out[i] = dist_point(xyz + 3 * i, xx[i], p, pp) - vdw[i];
}

/* max_dim (utilities.py:355-372): argmax over the upper triangle (diagonal
 * included, value 0 + 2 vdw_i) of d_ij + (vdw_i + vdw_j); first maximum in
 * row-major order.  N x N call shape: dgemm association, diagonal forced 0. */
/* The edge tile of OpenBLAS's dsyrk (SkylakeX kernels), established entry by entry against numpy's X @ X.T
 * (tests/tools/distance_order_probe.py --rule) and against sklearn on tests/golden/edge_tile.npz: when
 * N % 8 >= 4 and N <= 382 (from 383 the BLAS threads the product and the answer depends on the machine), an entry
 * between one of the atoms [8*(N/8), 8*(N/8)+4) and atom c is summed as fma(z,z', x*x' + y*y') when c is in the
 * first 12*floor(w/12) columns of the kernel call that covers it -- one call of width P for the columns left of
 * the last row panel (P = 0 up to N = 192, 32*ceil(floor(N/2)/32) above), one per 32 columns inside it
 * (w = min(32, N - 32*floor(c/32))); every other entry as fma(z,z', fma(y,y', x*x')). */
/* Where the last row panel starts: panels of 192 rows (GEMM_P) while at least 384 are left, then what is left in two
 * halves rounded to 32, then the rest -- the level-3 driver's recurrence with ONE BLAS thread (from 383 atoms OpenBLAS
 * threads the product and the reference's last bit follows the core count; the platform restated is one thread). */
static int64_t last_panel_start(int64_t n) {
    int64_t start = 0;
    for (;;) {
        int64_t rem = n - start;
        int64_t mi = rem >= 384 ? 192 : (rem > 192 ? 32 * ((rem / 2 + 31) / 32) : rem);
        if (start + mi >= n) return start;
        start += mi;
    }
}
static int edge_order(int64_t n, int64_t i, int64_t j) {
    if (n % 8 < 4) return 0;
    int64_t t0 = 8 * (n / 8);
    int ei = i >= t0 && i < t0 + 4, ej = j >= t0 && j < t0 + 4;
    if (!ei && !ej) return 0;
    int64_t c = ei ? j : i;
    int64_t panel = last_panel_start(n);
    if (c < panel) return c < 12 * (panel / 12);
    int64_t w = n - 32 * (c / 32);
    if (w > 32) w = 32;
    return (c % 32) < 12 * (w / 12);
}

double pwo_max_dim(int64_t n, const double *xyz, const double *xx, const double *vdw,
                   int64_t *oi, int64_t *oj) {
    double best = -INFINITY;
    int64_t bi = 0, bj = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *a = xyz + 3 * i;
        for (int64_t j = i; j < n; ++j) {
            const double *b = xyz + 3 * j;
            double d;
            if (i == j) {
                d = 0.0;
            } else {
                double g = edge_order(n, i, j) ? fma(a[2], b[2], a[0] * b[0] + a[1] * b[1])
                                               : fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]));
                double d2 = ((-2.0 * g) + xx[i]) + xx[j];
                d = sqrt(d2 > 0.0 ? d2 : 0.0);
            }
            double v = d + (vdw[i] + vdw[j]);
            if (v > best) { best = v; bi = i; bj = j; }
        }
    }
    /* np.triu zeroes the strict lower triangle: a 0 there can only win if every
     * upper-triangle entry is <= 0, impossible with positive radii. */
    *oi = bi; *oj = bj;
    return best;
}

/* ---- periodic pre-processing (oracle/pw_rebuild.py) ------------------------------------ */

/* distances of n atoms to one point (N x 1 call shape), utilities.py:984, 1001 */
void pwo_dists(int64_t n, const double *xyz, const double *xx, const double *p, double *out) {
    double pp = sqnorm3(p);
    for (int64_t i = 0; i < n; ++i) out[i] = dist_point(xyz + 3 * i, xx[i], p, pp);
}

/* numpy (3,3) matrix times (3,1) column as np.matrix.__mul__ evaluates it through
 * OpenBLAS (utilities.py:727-729, 738-740): y_i = fma(m_i2, x2, fma(m_i0, x0, m_i1 * x1));
 * established against numpy on random inputs (tests/test_rebuild.py). */
void pwo_mat3_apply(const double *m, int64_t n, const double *x, double *y) {
    for (int64_t k = 0; k < n; ++k) {
        const double *v = x + 3 * k;
        for (int i = 0; i < 3; ++i)
            y[3 * k + i] = fma(m[3 * i + 2], v[2], fma(m[3 * i + 0], v[0], m[3 * i + 1] * v[1]));
    }
}

/* Python's round(float, 8) (utilities.py:191-193, 203-205): the decimal expansion of the
 * double, correctly rounded (half-even) at 8 places, converted back correctly rounded.
 * p = x * 1e8 is inexact; fma gives its exact residual, which settles the tie cases.
 * Valid for |x| < 4.5e7 (2^52 / 1e8). */
double pwo_round8_one(double x) {
    double p = x * 1e8;
    if (!(fabs(p) < 4503599627370496.0)) return x;
    double e = fma(x, 1e8, -p);
    double f = floor(p);
    double r = p - f;
    double k;
    if (r < 0.5) k = f;
    else if (r > 0.5) k = f + 1.0;
    else if (e > 0.0) k = f + 1.0;
    else if (e < 0.0) k = f;
    else k = (fmod(f, 2.0) == 0.0) ? f : f + 1.0;
    /* r < 0.5 but p + e may still cross down to an integer boundary only when r == 0:
     * p integral and e < 0 means the true value is just below p: still rounds to p */
    return k / 1e8;
}
void pwo_round8(int64_t n, const double *in, double *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = pwo_round8_one(in[i]);
}
