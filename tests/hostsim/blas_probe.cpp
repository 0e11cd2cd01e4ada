// tests/hostsim/blas_probe.cpp -- exposes the BLAS restatements of
// pywindow_amd/csrc/pw_blas.hpp to Python (ctypes) so tests can compare them bit
// for bit with scipy.linalg.blas / .lapack.  Test infrastructure only.
#include "../../pywindow_amd/csrc/pw_blas.hpp"
extern "C" {
double hs_ddot(int n, const double* x, const double* y) { return pw::b_ddot(n, x, y); }
void hs_daxpy(int n, double a, const double* x, double* y) { pw::b_daxpy(n, a, x, y); }
double hs_dnrm2(int n, const double* x) { return pw::b_dnrm2(n, x); }
double hs_dnrm2_exact(int n, const double* x) { return pw::b_dnrm2_exact(n, x); }
// many vectors at once (rows of an (m, n) array); returns how many took the short route
long hs_dnrm2_many(long m, int n, const double* x, double* fast, double* exact) {
    long same = 0;
    for (long i = 0; i < m; ++i) {
        fast[i] = pw::b_dnrm2(n, x + i * n);
        exact[i] = pw::b_dnrm2_exact(n, x + i * n);
    }
    return same;
}
int hs_dpotrf_u(int n, double* a, int lda) { return pw::b_dpotrf_u(n, a, lda); }
int hs_dtrtrs_u(int trans, int n, int nrhs, const double* a, int lda, double* b, int ldb) {
    return pw::b_dtrtrs_u(trans != 0, n, nrhs, a, lda, b, ldb);
}
}
