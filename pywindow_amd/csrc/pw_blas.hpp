// pw_blas.hpp -- bit-faithful scalar restatements of the handful of BLAS/LAPACK
// routines SciPy 1.15's L-BFGS-B (scipy/optimize/__lbfgsb.c, a C translation of
// L-BFGS-B 3.0) calls, for the tiny shapes that occur on the pywindow hot path
// (n <= 3 variables, m = 10 corrections => vectors of length <= 20, triangular
// systems and Cholesky factors of order <= 20).
//
// Why this exists: the reference's opt_pore_diameter / window z-search
// (utilities.py:422, :1301) are scipy.optimize.minimize(method L-BFGS-B) runs
// whose outcome is chaotic at the ulp level, so matching
// the reference to 1e-6 needs the exact floating-point association of the
// OpenBLAS 0.3.28 kernels SciPy links (x86-64 AVX2/AVX-512 code paths).  Every
// routine below was established against scipy.linalg.blas / .lapack bit for
// bit (tests/test_blas_emulation.py repeats that check wherever SciPy is
// importable).
//
// All functions are plain loops on raw pointers: they compile for the host
// (tests/hostsim) and for gfx950 (pw_kernels.hip) from this one source.
#pragma once
#include "pw_common.hpp"
#include "pw_ext64.hpp"

namespace pw {

// ---- level 1 -------------------------------------------------------------
// ddot, unit stride, n < 32.  kernel/x86_64/ddot.c: the first n & -16 elements
// go through the AVX kernel (4 accumulators x 4 lanes, FMA), the tail is a
// sequential FMA chain onto the partial result.
PW_HD inline double b_ddot(int n, const double* x, const double* y) {
    double dot = 0.0;
    int i = 0;
    if (n >= 16) {
        double s[4];
        for (int l = 0; l < 4; ++l) {
            double a0 = x[l] * y[l];
            double a1 = x[4 + l] * y[4 + l];
            double a2 = x[8 + l] * y[8 + l];
            double a3 = x[12 + l] * y[12 + l];
            s[l] = ((a0 + a1) + a2) + a3;
        }
        dot = (s[0] + s[2]) + (s[1] + s[3]);
        i = 16;
    }
    for (; i < n; ++i) dot = pw_fma(y[i], x[i], dot);
    return dot;
}

// daxpy: y += a*x, one FMA per element.
PW_HD inline void b_daxpy(int n, double a, const double* x, double* y) {
    for (int i = 0; i < n; ++i) y[i] = pw_fma(a, x[i], y[i]);
}

PW_HD inline void b_dscal(int n, double a, double* x) {
    for (int i = 0; i < n; ++i) x[i] = a * x[i];
}

PW_HD inline void b_dcopy(int n, const double* x, double* y) {
    for (int i = 0; i < n; ++i) y[i] = x[i];
}

// dnrm2 (kernel/x86_64/nrm2.S): x87 code -- squares and running sum carried in
// 80-bit extended precision (64-bit significand), fsqrt in extended, result
// rounded to double on return.  Emulated with integers in pw_ext64.hpp (b_dnrm2).

// ---- dgemv 'T' as used inside dpotf2: y -= A^T x, A is m x n column-major ---
// kernel/x86_64/dgemv_t_4.c: rows in chunks of 4 (m & -4) through the
// 4x4 / 4x2 / 4x1 micro-kernels (columns grouped 4,2,1), the m & 3 leftover
// rows in a scalar tail.  Only m <= 11 is supported (m & -4 in {0,4,8}).
// One output element of that product: returns the NEW y_k given the old one.
// `k` is the column's position among the n columns (it selects the micro-kernel).
PW_HD inline double b_dgemv_t_elem(int m, int n, int k, const double* a, const double* x, double yk) {
    const int m1 = m & -4, m3 = m & 3;
    const int n4 = (n >> 2) << 2;
    if (m1) {
        double t;
        if (k < n4) {
            // 4x4 AVX2 kernel: one 4-lane FMA accumulator over the chunks
            double l0 = a[0] * x[0], l1 = a[1] * x[1], l2 = a[2] * x[2], l3 = a[3] * x[3];
            if (m1 == 8) {
                l0 = pw_fma(a[4], x[4], l0);
                l1 = pw_fma(a[5], x[5], l1);
                l2 = pw_fma(a[6], x[6], l2);
                l3 = pw_fma(a[7], x[7], l3);
            }
            t = (l0 + l2) + (l1 + l3);
        } else if ((n & 2) && k < n4 + 2) {
            // 4x2 SSE2 kernel: separately rounded products, 2-lane accumulator
            double q0 = a[0] * x[0] + a[2] * x[2];
            double q1 = a[1] * x[1] + a[3] * x[3];
            if (m1 == 8) {
                q0 = q0 + (a[4] * x[4] + a[6] * x[6]);
                q1 = q1 + (a[5] * x[5] + a[7] * x[7]);
            }
            t = q0 + q1;
        } else {
            // 4x1 kernel
            double l0 = a[0] * x[0], l1 = a[1] * x[1], l2 = a[2] * x[2], l3 = a[3] * x[3];
            if (m1 == 8) {
                l0 = l0 + a[4] * x[4];
                l1 = l1 + a[5] * x[5];
                l2 = l2 + a[6] * x[6];
                l3 = l3 + a[7] * x[7];
            }
            t = (l0 + l2) + (l1 + l3);
        }
        yk = yk - t;
    }
    if (m3 == 1) {
        yk = pw_fma(a[m1], -x[m1], yk);
    } else if (m3 == 2) {
        double t = a[m1 + 1] * (-x[m1 + 1]);
        t = pw_fma(a[m1], -x[m1], t);
        yk = yk + t;
    } else if (m3 == 3) {
        double t = a[m1 + 1] * (-x[m1 + 1]);
        t = pw_fma(a[m1], -x[m1], t);
        t = pw_fma(a[m1 + 2], -x[m1 + 2], t);
        yk = yk + t;
    }
    return yk;
}
PW_HD inline void b_dgemv_t_sub(int m, int n, const double* A, int lda, const double* x,
                                double* y, int incy) {
    for (int k = 0; k < n; ++k) {
        double* yk = y + (long)k * incy;
        *yk = b_dgemv_t_elem(m, n, k, A + (long)k * lda, x, *yk);
    }
}

// ---- dpotrf 'U', n <= 11 (lapack/potf2/potf2_U.c, unblocked) ----------------
// returns 0 or the 1-based index of the first non-positive pivot.
PW_HD inline int b_dpotrf_u(int n, double* a, int lda) {
    for (int j = 0; j < n; ++j) {
        double* cj = a + (long)j * lda;
        double ajj = cj[j] - b_ddot(j, cj, cj);
        if (ajj <= 0.0) {
            cj[j] = ajj;
            return j + 1;
        }
        ajj = pw_sqrt(ajj);
        cj[j] = ajj;
        int i = n - j - 1;
        if (i > 0) {
            b_dgemv_t_sub(j, i, a + (long)(j + 1) * lda, lda, cj, a + j + (long)(j + 1) * lda, lda);
            double inv = 1.0 / ajj;
            for (int k = 0; k < i; ++k) a[j + (long)(j + 1 + k) * lda] *= inv;
        }
    }
    return 0;
}


// ---- dtrtrs 'U', non-unit diagonal (interface/lapack/trtrs.c) -----------------
// One right-hand side goes through the level-2 TRSV drivers
// (driver/level2/trsv_U.c / trsv_L.c): true divisions by the diagonal, AXPY
// (no-trans) or DOT (trans) updates.  n <= 31.
PW_HD inline void b_dtrsv_un(int n, const double* a, int lda, double* x) {
    for (int i = n - 1; i >= 0; --i) {
        const double* ci = a + (long)i * lda;
        x[i] = x[i] / ci[i];
        double nx = -x[i];
        for (int k = 0; k < i; ++k) x[k] = pw_fma(nx, ci[k], x[k]);
    }
}
PW_HD inline void b_dtrsv_ut(int n, const double* a, int lda, double* x) {
    for (int i = 0; i < n; ++i) {
        const double* ci = a + (long)i * lda;
        if (i > 0) x[i] = x[i] - b_ddot(i, ci, x);
        x[i] = x[i] / ci[i];
    }
}
// Several right-hand sides ('U','T') go through the level-3 TRSM driver with the
// generic LT kernel (kernel/generic/trsm_kernel_LT.c, GEMM_UNROLL_M = 16):
// rows in blocks 16,8,4,2,1; per block a GEMM update with the rows already
// solved (FMA chain from zero, then one subtraction) followed by a
// right-looking solve that multiplies by the pre-inverted diagonal.  n <= 15
// here (no full 16-row block), which covers col <= m = 10.
// dinv: the reciprocals 1 / a(i, i) where the caller has them tabulated (the same quotients), else null.
PW_HD inline void b_dtrsm_ut_col(int n, const double* a, int lda, double* x, const double* dinv = nullptr) {
    int s = 0;
    for (int bs = 8; bs > 0; bs >>= 1) {
        if (!(n & bs)) continue;
        int e = s + bs;
        if (s > 0) {
            for (int k = s; k < e; ++k) {
                const double* ck = a + (long)k * lda;
                double acc = ck[0] * x[0];
                for (int j = 1; j < s; ++j) acc = pw_fma(ck[j], x[j], acc);
                x[k] = x[k] - acc;
            }
        }
        for (int i = s; i < e; ++i) {
            x[i] = x[i] * (dinv ? dinv[i] : 1.0 / a[i + (long)i * lda]);
            double nx = -x[i];
            for (int k = i + 1; k < e; ++k) x[k] = pw_fma(nx, a[i + (long)k * lda], x[k]);
        }
        s = e;
    }
}
PW_HD inline void b_dtrsm_ut(int n, int nrhs, const double* a, int lda, double* b, int ldb) {
    for (int c = 0; c < nrhs; ++c) b_dtrsm_ut_col(n, a, lda, b + (long)c * ldb);
}
// LAPACK dtrtrs front end: singularity check (exact zero on the diagonal).
PW_HD inline int b_dtrtrs_u(bool trans, int n, int nrhs, const double* a, int lda, double* b,
                            int ldb) {
    for (int i = 0; i < n; ++i)
        if (a[i + (long)i * lda] == 0.0) return i + 1;
    if (nrhs == 1) {
        if (trans) b_dtrsv_ut(n, a, lda, b);
        else b_dtrsv_un(n, a, lda, b);
    } else {
        // only the transposed multi-RHS form occurs (formk)
        b_dtrsm_ut(n, nrhs, a, lda, b, ldb);
    }
    return 0;
}

}  // namespace pw
