// pw_lbfgsb.hpp -- trajectory-faithful L-BFGS-B for n <= 3 variables, m = 10.
//
// The reference maximises the included-sphere diameter with
// scipy.optimize.minimize(..., bounds=...) (utilities.py:422 for the pore
// centre, n = 3; utilities.py:1301 for the window neck along z, n = 1).  With
// bounds and no method SciPy picks L-BFGS-B (scipy/optimize/_minimize.py),
// whose engine in SciPy 1.15 is a C translation of L-BFGS-B 3.0 (Zhu, Byrd,
// Lu, Nocedal; Morales & Nocedal 2011) driven by reverse communication from
// scipy/optimize/_lbfgsb_py.py:427-456.  Because the objective's optimum sits
// on a kink, the iterates are decided by last-bit comparisons;
// this file therefore follows the published algorithm statement by statement
// and performs every inner product, triangular solve and Cholesky
// factorisation with the association of the OpenBLAS kernels SciPy links
// (pw_blas.hpp).  It was validated by running SciPy's own setulb() in lockstep
// and comparing x, the workspace matrices and the task code after every call
// (tests/test_lbfgsb_lockstep.py).
//
// Execution model: the routines are written for a *team of lanes* (pw_team.hpp).
// Scalars and the length-n vectors (n <= 3) are computed redundantly by every
// lane; everything indexed by the correction count (vectors of length col / 2col,
// the col x col and 2col x 2col matrices) is spread over the lanes element by
// element, each element keeping the reference's own sequential accumulation
// order, with a wave-level sync between dependent phases.  Triangular solves run
// as systolic sweeps (one division per step, the updates of a step in parallel).
// With a one-lane team (host tests) the same code degenerates to the plain
// sequential algorithm.
//
// Written from the algorithm description; single-source for host tests and
// gfx950.  State lives in one plain struct (about 10 KB) that the HIP kernel
// keeps in LDS, one instance per wavefront.
#pragma once
#include "pw_blas.hpp"
#include "pw_team.hpp"

#if defined(PW_PROFILE) && defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_LB_TIMERS)
#define LB_T0(var) long long var = wall_clock64()
#define LB_T1(slot, var) do { if (prof && T::lane() == 0) atomicAdd(&prof[slot], (unsigned long long)(wall_clock64() - var)); } while (0)
#else
#define LB_T0(var) do {} while (0)
#define LB_T1(slot, var) do {} while (0)
#endif
// -DPW_PROFILE -DPW_LB_FINE: shader-clock timers of the optimiser's sub-phases in the slots the window search
// uses otherwise (run the chains' stages only: tests/tools/profile_chains.py)
#if defined(PW_PROFILE) && defined(PW_LB_FINE) && defined(__HIP_DEVICE_COMPILE__)
#define LB_F0(var) long long var = clock64()
#define LB_F1(slot, var) do { if (prof && T::lane() == 0) atomicAdd(&prof[slot], (unsigned long long)(clock64() - var)); } while (0)
#define LB_M0(var) long long var = clock64()
#define LB_M1(slot, var) do { if (m->prof_fine && T::lane() == 0) atomicAdd(&m->prof_fine[slot], (unsigned long long)(clock64() - var)); } while (0)
#else
#define LB_F0(var) do {} while (0)
#define LB_F1(slot, var) do {} while (0)
#define LB_M0(var) do {} while (0)
#define LB_M1(slot, var) do {} while (0)
#endif

namespace pw {

enum LbTask : int {
    LB_START = 0,
    LB_NEW_X = 1,
    LB_FG = 3,          // caller must supply f and g at x
    LB_CONVERGENCE = 4,
    LB_STOP = 5,
    LB_WARNING = 6,
    LB_ERROR = 7,
    LB_ABNORMAL = 8,
};
enum LbMsg : int {
    LBM_NONE = 0,
    LBM_FG_START = 301,
    LBM_FG_LNSRCH = 302,
    LBM_CONV_PGTOL = 401,
    LBM_CONV_FTOL = 402,
};

constexpr int LB_M = 10;

// State of the line search (dcsrch) between its calls.  It lives with the arrays, not in the optimiser object: the
// search is entered once per function evaluation, reads all of this in one batch, and nothing else ever looks at it --
// as members of the object these thirteen doubles were live (mostly spilled) across the whole optimiser stage.
struct LsState {
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
    int brackt, stage;
};

// The arrays of one optimiser instance: this block lives in team-shared memory (LDS).
template <int N>
struct LbMem {
    static constexpr int M = LB_M;
    static constexpr int M2 = 2 * LB_M;
    double l[N], u[N];
    double x[N], g[N];
    double ws[M * N], wy[M * N];
    double sy[M * M], ss[M * M], wt[M * M];
    // WN (the factorised middle matrix) only ever uses its upper triangle and WN1 (the running
    // inner products it is rebuilt from) only its lower one: they share one 2m x 2m array; the
    // diagonal of WN1 lives in wn1d
    double wn[M2 * M2], wn1d[M2];
    double z[N], r[N], d[N], t[N], xp[N];
    double wa[8 * M];
    double acc[2 * M];  // running dot products of the systolic triangular solves
    // what the solves divide by, and the refined hardware reciprocal of each (pw_recip_hw): the diagonal of SY, its
    // square root, the diagonals of the factors WT and WN.  Device teams only (tables()); rebuilt wherever the
    // matrix is (matupd, formt, formk).
    double dsy[M], rsy[M], sqy[M], rsq[M], rwt[M], rwn[M2];
    double iwn[M];      // 1 / WN(i, i), i < col: the quotients potf2 scales its rows with, which trsm's solve multiplies by
    // bmv's two triangular sums, one (row, column) pair per lane: the quotients of a sum wait here for the lane that
    // adds them up in the reference's order; pair[lane] = row | column << 8 of lane's pair (row > column; 255: none)
    double tri[64];     // (also the col x nsub products of subsm's last step: two kinds, at [term] and [32 + term])
    int pair[64];
    double le[N], ue[N];    // the bounds a line-search iterate is put back on: l / u where there is one, -inf / +inf else
    LsState ls;
#if defined(PW_PROFILE) && defined(PW_LB_FINE)
    unsigned long long* prof_fine;  // sub-phase timers of the out-of-line wave routines (diagnostic builds only)
#endif
    int cand[32];       // atoms that can hold the minimum near the current reference point (pw_unit.hpp: NearGap4)
    int nbd[N];
    int index[N], iwhere[N], indx2[N];
};

// The optimiser object itself holds only scalars and pointers into its LbMem block; the
// kernels keep it in a local variable, i.e. in registers, so the line search does not pay
// an LDS round trip for every scalar it touches.
// ---- triangular solves of one wave with the unknowns in registers ------------------------------
// One element of x per lane; the pivot travels by v_readlane, so a step costs a division and an
// fma instead of an LDS round trip and a barrier.  The same operations are applied to every
// element, in the same order, as in the LDS sweeps of Lbfgsb::p_dtrsv_un / p_dtrsv_ut.  Out of
// line on purpose: inlined into the optimiser step they raise its register pressure enough to
// spill inside the hot loops.
template <class T>
PW_NOINLINE PW_HD inline void lb_trsv_un_wave(int n, const double* a, int lda, double* x) {
    PW_ASSUME_LDS(a);
    PW_ASSUME_LDS(x);
    const int lane = T::lane();
    double xk = lane < n ? x[lane] : 0.0;
    for (int i = n - 1; i >= 0; --i) {
        const double* ci = a + (long)i * lda;
        double xi = T::bcast_u(xk, i) / ci[i];
        if (lane == i) xk = xi;
        if (lane < i) xk = pw_fma(-xi, ci[lane], xk);
    }
    T::wave_sync();
    if (lane < n) x[lane] = xk;
    T::wave_sync();
}
template <class T>
PW_NOINLINE PW_HD inline void lb_trsv_ut_wave(int n, const double* a, int lda, double* x) {
    PW_ASSUME_LDS(a);
    PW_ASSUME_LDS(x);
    const int lane = T::lane();
    const double* ci = a + (long)(lane < n ? lane : 0) * lda;     // my row of U^T = column of U
    double xk = lane < n ? x[lane] : 0.0;
    double ak = 0.0;
    for (int s = 0; s < n; ++s) {
        double xs = T::bcast_u(xk, s);
        if (s > 0) xs = xs - T::bcast_u(ak, s);
        xs = xs / a[s + (long)s * lda];
        if (lane == s) xk = xs;
        if (lane > s && lane < n) {
            if (lane >= 16 && s < 16) {
                // rows 16.. take their first 16 terms through the SIMD ddot kernel order
                if (s == 15) {
                    double sl[4];
                    for (int l = 0; l < 4; ++l) {
                        double a0 = ci[l] * T::bcast_u(xk, l);
                        double a1 = ci[4 + l] * T::bcast_u(xk, 4 + l);
                        double a2 = ci[8 + l] * T::bcast_u(xk, 8 + l);
                        double a3 = ci[12 + l] * T::bcast_u(xk, 12 + l);
                        sl[l] = ((a0 + a1) + a2) + a3;
                    }
                    ak = (sl[0] + sl[2]) + (sl[1] + sl[3]);
                }
            } else {
                ak = pw_fma(xs, ci[s], ak);
            }
        }
    }
    T::wave_sync();
    if (lane < n) x[lane] = xk;
    T::wave_sync();
}


#if defined(__HIP_DEVICE_COMPILE__)
// the out-of-line wave routines below ask a team for its lane number, lane-to-lane moves and the wave-level ordering
// of LDS traffic only -- the same for every team of whole waves, so they are instantiated once, for this one
typedef DeviceTeam<1> LbWave;
// arguments of an out-of-line function arrive in vector registers: tell the compiler which are the same in every lane
__device__ inline int lb_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class P>
__device__ inline P* lb_uniform(P* p) {
    union { P* p; int i[2]; } a;
    a.p = p;
    a.i[0] = __builtin_amdgcn_readfirstlane(a.i[0]);
    a.i[1] = __builtin_amdgcn_readfirstlane(a.i[1]);
    return a.p;
}
// ---- one wave, everything in registers ------------------------------------------------------------------------
// The solves below are the ones of lb_trsv_ut_wave / lb_trsv_un_wave with (a) the lane's column (row) of the
// factor and its diagonal loaded ONCE, before the sweep, instead of two LDS round trips per step, and (b) the
// division of a step done by every lane on its own element with the tabulated reciprocal (pw_div_r: three
// dependent instructions, the bits of the true quotient) -- the lane whose turn it is holds the finished numerator,
// the others compute a value nobody reads.  Same operations on every element, in the same order.
// SPEC: the divisions without their guard (pw_div_ru); *tnum receives the lane's own dividend, which the caller
// checks afterwards (lb_plain) -- if any was zero or outside the plain range the caller repeats the whole computation
// with SPEC = false, where every division is guarded.
template <class T, int NMAX, bool SPEC>
__device__ inline __attribute__((always_inline)) double lb_solve_ut_reg(int n, const double* a, int lda, const double* rdiag,
                                                                       double xk, double* tnum) {
    PW_ASSUME_LDS(a);
    PW_ASSUME_LDS(rdiag);
    const int lane = T::lane();
    const bool act = lane < n;
    const int li = act ? lane : 0;
    const double* ci = a + li * lda;              // my row of U^T = column of U
    double c[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) c[k] = ci[k];
    int dli = li * (lda + 1);
    asm volatile("" : "+v"(dli));                 // (a load of its own: not an indexed read of c[])
    double dg = a[dli], rd = rdiag[li];
    if (!act) { dg = 1.0; rd = 1.0; xk = 0.0; }
    const double x0 = xk;
    double ak = 0.0;
#pragma unroll
    for (int s = 0; s < NMAX; ++s) {
        if (s >= n) continue;                     // (not a break: the trip count stays a constant, the loop unrolls)
        const double mine = SPEC ? pw_div_ru(xk - ak, dg, rd) : pw_div_r(xk - ak, dg, rd);
        const double xs = T::bcast_u(mine, s);
        xk = lane == s ? xs : xk;
        if (NMAX > 16 && s < 16) {
            // rows 16.. take their first 16 terms through the SIMD ddot kernel order
            if (s == 15) {
                double sl[4];
#pragma unroll
                for (int l = 0; l < 4; ++l) {
                    double a0 = c[l] * T::bcast_u(xk, l);
                    double a1 = c[4 + l] * T::bcast_u(xk, 4 + l);
                    double a2 = c[8 + l] * T::bcast_u(xk, 8 + l);
                    double a3 = c[12 + l] * T::bcast_u(xk, 12 + l);
                    sl[l] = ((a0 + a1) + a2) + a3;
                }
                const double blk = (sl[0] + sl[2]) + (sl[1] + sl[3]);
                ak = (act && lane >= 16) ? blk : ak;
            }
            ak = (act && lane > s && lane < 16) ? pw_fma(xs, c[s], ak) : ak;
        } else {
            ak = (act && lane > s) ? pw_fma(xs, c[s], ak) : ak;
        }
    }
    if (SPEC) *tnum = act ? x0 - ak : 1.0;        // what this lane divided when its turn came
    return xk;
}
template <class T, int NMAX, bool SPEC>
__device__ inline __attribute__((always_inline)) double lb_solve_un_reg(int n, const double* a, int lda, const double* rdiag,
                                                                       double xk, double* tnum) {
    PW_ASSUME_LDS(a);
    PW_ASSUME_LDS(rdiag);
    const int lane = T::lane();
    const bool act = lane < n;
    const int li = act ? lane : 0;
    double rw[NMAX];                               // my row of U
#pragma unroll
    for (int i = 0; i < NMAX; ++i) rw[i] = a[li + i * lda];
    int dli = li * (lda + 1);
    asm volatile("" : "+v"(dli));
    double dg = a[dli], rd = rdiag[li];
    if (!act) { dg = 1.0; rd = 1.0; xk = 0.0; }
    double tn = 1.0;
#pragma unroll
    for (int i = NMAX - 1; i >= 0; --i) {
        if (i >= n) continue;
        const double mine = SPEC ? pw_div_ru(xk, dg, rd) : pw_div_r(xk, dg, rd);
        const double xi = T::bcast_u(mine, i);
        if (SPEC) tn = lane == i ? xk : tn;
        xk = lane == i ? xi : xk;
        xk = lane < i ? pw_fma(-xi, rw[i], xk) : xk;
    }
    if (SPEC) *tnum = tn;
    return xk;
}
// the biased exponent field of x, and the test "every dividend between emin and emax was plain" (pw_plain_exponent)
__device__ inline unsigned lb_expo(double x) {
    union { double d; unsigned long long u; } c;
    c.d = x;
    return (unsigned)(c.u >> 32) & 0x7ff00000u;
}
__device__ inline bool lb_plain_range(unsigned emin, unsigned emax) {
    return (emin - 0x2bc00000u) <= 0x2bc00000u && (emax - 0x2bc00000u) <= 0x2bc00000u;
}

// bmv for one wave: lane i owns p[i] and p[col + i]; reads everything, then writes (Lbfgsb::bmv has the statement).
// Returns the singularity code of dtrtrs (0: fine; non-zero: nothing written), or -1 when SPEC and a dividend was not
// plain: p may hold part of a result then, and the guarded routine the caller takes next rewrites all of it.
template <class T, int N, bool SPEC>
__device__ inline __attribute__((always_inline)) int lb_bmv_body(LbMem<N>* m, int col, const double* v, double* p) {
    constexpr int M = LB_M;
    const int lane = T::lane();
    const bool act = lane < col;
    const int li = act ? lane : 0;
    {
        const unsigned long long z = T::ballot(act && m->wt[li + M * li] == 0.0);      // dtrtrs' singularity test
        if (z) return (int)__builtin_ctzll(z) + 1;
    }
    LB_M0(tb0);
    const double vi = act ? v[li] : 0.0, vc = act ? v[col + li] : 0.0;
    unsigned emin = 0x3ff00000u, emax = 0x3ff00000u;      // exponent range of the dividends (SPEC)
    // The two triangular sums, one quotient per lane: lane q owns the pair (a, b), a > b (LbMem::pair), i.e. the entry
    // SY(a, b) and the divisor SY(b, b) of BOTH sums -- the first wants SY(a, b) v[b] / SY(b, b) in row a, the second
    // SY(a, b) p[col + a] / SY(b, b) in row b.  The quotients go through LbMem::tri to the lane that owns the row,
    // which adds them in the reference's order (column ascending / row ascending, from +0).
    const int pk = m->pair[lane];
    const int pa_ = pk & 0xff, pb_ = (pk >> 8) & 0xff;
    const bool pact = pa_ < col;
    const int sa = pact ? pa_ : 1, sb = pact ? pb_ : 0;
    const double sy_ab = m->sy[sa + M * sb], d_b = m->dsy[sb], r_b = m->rsy[sb];
    // p[col + i] = v[col + i] + sum_{k < i} SY(i, k) * v[k] / SY(k, k)
    {
        const double num = pact ? sy_ab * v[sb] : 1.0;
        if (SPEC) {
            const unsigned e = lb_expo(num);
            emin = e < emin ? e : emin;
            emax = e > emax ? e : emax;
        }
        const double term = SPEC ? pw_div_ru(num, d_b, r_b) : pw_div_r(num, d_b, r_b);
        m->tri[lane] = term;
    }
    T::wave_sync();
    double sum = 0.0;
    {
        // (straight-line: every read first -- a branch per term would make each a round trip of its own)
        const int base = li * (li - 1) / 2;
        double q[M - 1];
#pragma unroll
        for (int k = 0; k < M - 1; ++k) q[k] = m->tri[(base + k) & 63];
#pragma unroll
        for (int k = 0; k < M - 1; ++k) {
            const bool on = act && k < lane && k + 1 < col;
            sum = sum + (on ? q[k] : 0.0);          // (+0 leaves a sum that started from +0 as it is)
        }
    }
    double xk = lane == 0 ? vc : vc + sum;
    double t_ut = 1.0, t_un = 1.0;
    LB_M1(3, tb0);
    LB_M0(tb1);
    xk = lb_solve_ut_reg<T, M, SPEC>(col, m->wt, M, m->rwt, xk, &t_ut);
    LB_M1(4, tb1);
    LB_M0(tb2);
    const double sq = act ? m->sqy[li] : 1.0, rs = act ? m->rsq[li] : 1.0;
    const double pa = SPEC ? pw_div_ru(vi, sq, rs) : pw_div_r(vi, sq, rs);
    xk = lb_solve_un_reg<T, M, SPEC>(col, m->wt, M, m->rwt, xk, &t_un);
    LB_M1(5, tb2);
    LB_M0(tb3);
    // p[i] = -p[i] / sqrt(SY(i, i)) + sum_{k > i} SY(k, i) * p[col + k] / SY(i, i)
    const double pi = SPEC ? pw_div_ru(-pa, sq, rs) : pw_div_r(-pa, sq, rs);
    if (act) p[col + lane] = xk;                // (the second half of the result; the pairs read it from here)
    T::wave_sync();
    {
        const double num = pact ? sy_ab * p[col + sa] : 1.0;
        if (SPEC) {
            const unsigned e = lb_expo(num);
            emin = e < emin ? e : emin;
            emax = e > emax ? e : emax;
        }
        const double term = SPEC ? pw_div_ru(num, d_b, r_b) : pw_div_r(num, d_b, r_b);
        m->tri[lane] = term;
    }
    T::wave_sync();
    double sum2 = 0.0;
    {
        double q[M];
#pragma unroll
        for (int k = 1; k < M; ++k) q[k] = m->tri[(k * (k - 1) / 2 + li) & 63];
#pragma unroll
        for (int k = 1; k < M; ++k) {
            const bool on = act && k > lane && k < col;
            sum2 = sum2 + (on ? q[k] : 0.0);
        }
    }
    if (SPEC) {
        // every dividend of this lane plain and non-zero, every tabulated reciprocal a real one
        unsigned e;
        e = lb_expo(t_ut); emin = e < emin ? e : emin; emax = e > emax ? e : emax;
        e = lb_expo(t_un); emin = e < emin ? e : emin; emax = e > emax ? e : emax;
        if (act) {
            e = lb_expo(vi); emin = e < emin ? e : emin; emax = e > emax ? e : emax;
            e = lb_expo(pa); emin = e < emin ? e : emin; emax = e > emax ? e : emax;
        }
        bool good = lb_plain_range(emin, emax) && rs != 0.0 && (!pact || r_b != 0.0) && (!act || m->rwt[li] != 0.0);
#ifdef PW_LB_FORCE_FALLBACK
        good = false;                     // (test builds: every call takes the guarded path after the speculative one)
#endif
        if (T::ballot(!good)) return -1;
    }
    if (act) p[lane] = pi + sum2;
    T::wave_sync();
    LB_M1(6, tb3);
    return 0;
}
template <class T, int N>
PW_NOINLINE __device__ inline int lb_bmv_wave_guarded(LbMem<N>* m, int col, const double* v, double* p) {
    m = lb_uniform(m); v = lb_uniform(v); p = lb_uniform(p); col = lb_uniform(col);
    PW_ASSUME_LDS(m);
    PW_ASSUME_LDS(v);
    PW_ASSUME_LDS(p);
    return lb_bmv_body<T, N, false>(m, col, v, p);
}
template <class T, int N>
PW_NOINLINE __device__ inline int lb_bmv_wave(LbMem<N>* m, int col, const double* v, double* p) {
    m = lb_uniform(m); v = lb_uniform(v); p = lb_uniform(p); col = lb_uniform(col);
    PW_ASSUME_LDS(m);
    PW_ASSUME_LDS(v);
    PW_ASSUME_LDS(p);
    const int r = lb_bmv_body<T, N, true>(m, col, v, p);
    if (r >= 0) return r;
    return lb_bmv_wave_guarded<T, N>(m, col, v, p);
}
// the two solves of subsm on the factor WN (order 2 col <= 20): wv <- WN^-1 diag(-I, I) WN^-T wv
template <class T, int N, bool SPEC>
__device__ inline __attribute__((always_inline)) int lb_subsm_solves_body(LbMem<N>* m, int col, double* wv) {
    constexpr int M2 = 2 * LB_M;
    const int lane = T::lane();
    const int n = 2 * col;
    const bool act = lane < n;
    const int li = act ? lane : 0;
    {
        const unsigned long long z = T::ballot(act && m->wn[li + M2 * li] == 0.0);
        if (z) return (int)__builtin_ctzll(z) + 1;
    }
    double xk = act ? wv[li] : 0.0;
    double t_ut = 1.0, t_un = 1.0;
    xk = lb_solve_ut_reg<T, M2, SPEC>(n, m->wn, M2, m->rwn, xk, &t_ut);
    xk = lane < col ? -xk : xk;
    xk = lb_solve_un_reg<T, M2, SPEC>(n, m->wn, M2, m->rwn, xk, &t_un);
    if (SPEC) {
        const unsigned e1 = lb_expo(t_ut), e2 = lb_expo(t_un);
        bool good = lb_plain_range(e1 < e2 ? e1 : e2, e1 > e2 ? e1 : e2) && (!act || m->rwn[li] != 0.0);
#ifdef PW_LB_FORCE_FALLBACK
        good = false;
#endif
        if (T::ballot(!good)) return -1;
    }
    if (act) wv[lane] = xk;
    T::wave_sync();
    return 0;
}
template <class T, int N>
PW_NOINLINE __device__ inline int lb_subsm_solves_wave_guarded(LbMem<N>* m, int col, double* wv) {
    m = lb_uniform(m); wv = lb_uniform(wv); col = lb_uniform(col);
    PW_ASSUME_LDS(m);
    PW_ASSUME_LDS(wv);
    return lb_subsm_solves_body<T, N, false>(m, col, wv);
}
template <class T, int N>
PW_NOINLINE __device__ inline int lb_subsm_solves_wave(LbMem<N>* m, int col, double* wv) {
    m = lb_uniform(m); wv = lb_uniform(wv); col = lb_uniform(col);
    PW_ASSUME_LDS(m);
    PW_ASSUME_LDS(wv);
    const int r = lb_subsm_solves_body<T, N, true>(m, col, wv);
    if (r >= 0) return r;
    return lb_subsm_solves_wave_guarded<T, N>(m, col, wv);
}
// ---- Cholesky factorisation of order <= 10 by one wave, the matrix in registers --------------------------------
// potf2 'U' as b_dpotrf_u / Lbfgsb::p_dpotrf_u do it (lapack/potf2/potf2_U.c with the dgemv_t micro-kernels of
// pw_blas.hpp), lane k holding column k of the upper triangle: the pivot's column travels by v_readlane, a column's
// running inner product (b_ddot's sequential FMA chain, i ascending) is kept up to date as its entries become final,
// and nothing goes through team memory between the first load and the last store -- the LDS form pays two round
// trips and two wave-level syncs per column.  For n <= 10 the three dgemv_t micro-kernels give the same bits where
// they can occur: with four rows done (m1 == 4) 4x4, 4x2 and 4x1 all compute (l0 + l2) + (l1 + l3) of separately
// rounded products, and with eight (m1 == 8, pivots 8 and 9) at most one column is left, which is the 4x1 kernel.
// Also leaves the tables the solves want: rtab[j] = pw_recip_hw(U(j, j)) and, if asked for, itab[j] = 1 / U(j, j)
// (the quotient potf2 itself scales the row with).  Returns 0, or -1 when a pivot was not positive: nothing has been
// written then and the caller takes the team-memory routine, which leaves what LAPACK leaves.
template <class T, int N>
PW_NOINLINE __device__ inline int lb_potrf_wave(LbMem<N>* m, double* a, int lda, int n, double* rtab, double* itab) {
    constexpr int NMAX = LB_M;
    m = lb_uniform(m); a = lb_uniform(a); rtab = lb_uniform(rtab); itab = lb_uniform(itab);
    lda = lb_uniform(lda); n = lb_uniform(n);
    PW_ASSUME_LDS(m);
    PW_ASSUME_LDS(a);
    PW_ASSUME_LDS(rtab);
    const int lane = T::lane();
    const bool act = lane < n;
    const int li = act ? lane : 0;
    double c[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) c[i] = a[i + li * lda];      // (rows below the diagonal: never used, never stored)
    double dsum = 0.0;          // b_ddot(j, column, column) of MY column, as far as its entries are final
    double inv_mine = 1.0, piv_mine = 1.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j >= n) continue;
        // pivot: lane j's c[j] - ddot(j, cj, cj)
        const double ajj = T::bcast_u(c[j] - dsum, j);
        bad = bad || !(ajj > 0.0);
        const double sq = pw_sqrt(ajj);
        const double inv = 1.0 / sq;
        // the pivot column's finished entries, for the product with the columns to its right
        double x[NMAX];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) x[i] = i < j ? T::bcast_u(c[i], j) : 0.0;
        // b_dgemv_t_elem(m = j, ...) on my column: y = c[j] - sum_i c[i] x[i] in the micro-kernels' association
        double yk = c[j];
        const int m1 = j & -4, m3 = j & 3;
        if (m1) {
            double l0 = c[0] * x[0], l1 = c[1] * x[1], l2 = c[2] * x[2], l3 = c[3] * x[3];
            if (m1 == 8) {
                l0 = l0 + c[4] * x[4];
                l1 = l1 + c[5] * x[5];
                l2 = l2 + c[6] * x[6];
                l3 = l3 + c[7] * x[7];
            }
            yk = yk - ((l0 + l2) + (l1 + l3));
        }
        if (m3 == 1) {
            yk = pw_fma(c[m1], -x[m1], yk);
        } else if (m3 == 2) {
            double t = c[m1 + 1] * (-x[m1 + 1]);
            t = pw_fma(c[m1], -x[m1], t);
            yk = yk + t;
        } else if (m3 == 3) {
            double t = c[m1 + 1] * (-x[m1 + 1]);
            t = pw_fma(c[m1], -x[m1], t);
            t = pw_fma(c[m1 + 2], -x[m1 + 2], t);
            yk = yk + t;
        }
        const double cj = lane > j ? yk * inv : (lane == j ? sq : c[j]);
        c[j] = cj;
        dsum = lane > j ? pw_fma(cj, cj, dsum) : dsum;
        inv_mine = lane == j ? inv : inv_mine;
        piv_mine = lane == j ? sq : piv_mine;
    }
    if (T::ballot(bad) != 0ull) return -1;
    if (act) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if (i <= lane) a[i + lane * lda] = c[i];
        rtab[lane] = pw_recip_hw(piv_mine);
        if (itab) itab[lane] = inv_mine;
    }
    T::wave_sync();
    return 0;
}
#endif

// ---- dcstep (More'-Thuente safeguarded step) -----------------------------------
PW_HD inline void lb_dcstep(double& stx, double& fx, double& dx, double& sty, double& fy,
                         double& dy, double& stp, double fp, double dp, int& brackt,
                         double stpmin, double stpmax) {
    double gamma, p, q, rr, s, sgnd, stpc, stpf, stpq, th;
    // dx / |dx| is +-1 exactly for every finite non-zero dx (0 / 0 and inf / inf: the division itself, a NaN)
    {
        const double adx = pw_abs(dx);
        double unit = dx < 0.0 ? -1.0 : 1.0;
        if (__builtin_expect(!(adx > 0.0 && adx < PW_INF), 0)) unit = dx / adx;
        sgnd = dp * unit;
    }
    if (fp > fx) {
        th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = pw_max(pw_max(pw_abs(th), pw_abs(dx)), pw_abs(dp));
        gamma = s * pw_sqrt((th / s) * (th / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        p = (gamma - dx) + th;
        q = ((gamma - dx) + gamma) + dp;
        rr = p / q;
        stpc = stx + rr * (stp - stx);
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        if (pw_abs(stpc - stx) < pw_abs(stpq - stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        brackt = true;
    } else if (sgnd < 0.0) {
        th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = pw_max(pw_max(pw_abs(th), pw_abs(dx)), pw_abs(dp));
        gamma = s * pw_sqrt((th / s) * (th / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + th;
        q = ((gamma - dp) + gamma) + dx;
        rr = p / q;
        stpc = stp + rr * (stx - stp);
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (pw_abs(stpc - stp) > pw_abs(stpq - stp)) stpf = stpc;
        else stpf = stpq;
        brackt = true;
    } else if (pw_abs(dp) < pw_abs(dx)) {
        th = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = pw_max(pw_max(pw_abs(th), pw_abs(dx)), pw_abs(dp));
        gamma = s * pw_sqrt(pw_max(0.0, (th / s) * (th / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + th;
        q = (gamma + (dx - dp)) + gamma;
        rr = p / q;
        if (rr < 0.0 && gamma != 0.0) stpc = stp + rr * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            if (pw_abs(stpc - stp) < pw_abs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            if (stp > stx) stpf = pw_min(stp + 0.66 * (sty - stp), stpf);
            else stpf = pw_max(stp + 0.66 * (sty - stp), stpf);
        } else {
            if (pw_abs(stpc - stp) > pw_abs(stpq - stp)) stpf = stpc;
            else stpf = stpq;
            stpf = pw_min(stpmax, stpf);
            stpf = pw_max(stpmin, stpf);
        }
    } else {
        if (brackt) {
            th = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            s = pw_max(pw_max(pw_abs(th), pw_abs(dy)), pw_abs(dp));
            gamma = s * pw_sqrt((th / s) * (th / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            p = (gamma - dp) + th;
            q = ((gamma - dp) + gamma) + dy;
            rr = p / q;
            stpc = stp + rr * (sty - stp);
            stpf = stpc;
        } else if (stp > stx) {
            stpf = stpmax;
        } else {
            stpf = stpmin;
        }
    }
    // the interval update, as selects on values: written as conditional stores through the reference parameters the
    // compiler turned it into stores through a selected POINTER, which kept the whole line-search state in scratch
    // memory on the GPU
    {
        const bool hi = fp > fx;
        const bool sw = !hi && sgnd < 0.0;
        const double nsty = hi ? stp : (sw ? stx : sty);
        const double nfy = hi ? fp : (sw ? fx : fy);
        const double ndy = hi ? dp : (sw ? dx : dy);
        const double nstx = hi ? stx : stp;
        const double nfx = hi ? fx : fp;
        const double ndx = hi ? dx : dp;
        sty = nsty; fy = nfy; dy = ndy;
        stx = nstx; fx = nfx; dx = ndx;
    }
    stp = stpf;
}

// ---- dcsrch ------------------------------------------------------------------------
// One call of the line search.  Out of line, and its state in team memory (LsState): a scalar routine of a dozen
// divisions and a square root whose thirteen doubles of state were live -- spilled -- across the whole optimiser stage
// while it sat inside it; on its own it has its own registers.  ls_task in, {step, ls_task} out.
struct LsOut { double st; int task; };
// The search proper, on a state the caller holds (registers, for the fused line search of Lbfgsb::linesearch; a copy
// of the team-memory block for the reverse-communication form below).  ls_task in, {step, ls_task} out.
PW_HD inline __attribute__((always_inline)) LsOut lb_dcsrch_core(LsState& L, int ls_task, double fv, double gv, double st,
                                                                 double ftol, double gtol, double xtol, double stpmin,
                                                                 double stpmax) {
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    if (ls_task == 0) {
        if (st < stpmin) ls_task = 4;
        if (st > stpmax) ls_task = 4;
        if (gv >= 0.0) ls_task = 4;
        if (ls_task == 4) return LsOut{st, ls_task};
        L.brackt = 0;
        L.stage = 1;
        L.finit = fv;
        L.ginit = gv;
        L.gtest = ftol * L.ginit;
        L.width = stpmax - stpmin;
        L.width1 = L.width / p5;
        L.stx = 0.0; L.fx = L.finit; L.gx = L.ginit;
        L.sty = 0.0; L.fy = L.finit; L.gy = L.ginit;
        L.stmin = 0.0;
        L.stmax = st + xtrapu * st;
        return LsOut{st, 1};
    }
    double ftest = L.finit + st * L.gtest;
    if (L.stage == 1 && fv <= ftest && gv >= 0.0) L.stage = 2;
    if (L.brackt && (st <= L.stmin || st >= L.stmax)) ls_task = 3;
    if (L.brackt && L.stmax - L.stmin <= xtol * L.stmax) ls_task = 3;
    if (st == stpmax && fv <= ftest && gv <= L.gtest) ls_task = 3;
    if (st == stpmin && (fv > ftest || gv >= L.gtest)) ls_task = 3;
    if (fv <= ftest && pw_abs(gv) <= gtol * (-L.ginit)) ls_task = 2;
    if (ls_task == 3 || ls_task == 2) return LsOut{st, ls_task};
    {
        // dcstep on the function itself or, in the first stage while the sufficient-decrease test fails, on the
        // modified function (minus gtest * step).  ONE call either way, its operands chosen by selects: with a call
        // in each branch the compiler merged the two inlined copies behind pointers to the operands, and the state
        // they pointed to lived in scratch memory.
        const bool mod = L.stage == 1 && fv <= L.fx && fv > ftest;
        double fp = mod ? fv - st * L.gtest : fv;
        double fxw = mod ? L.fx - L.stx * L.gtest : L.fx;
        double fyw = mod ? L.fy - L.sty * L.gtest : L.fy;
        double dp = mod ? gv - L.gtest : gv;
        double gxw = mod ? L.gx - L.gtest : L.gx;
        double gyw = mod ? L.gy - L.gtest : L.gy;
        double stx = L.stx, sty = L.sty;
        int brackt = L.brackt;
        lb_dcstep(stx, fxw, gxw, sty, fyw, gyw, st, fp, dp, brackt, L.stmin, L.stmax);
        L.stx = stx; L.sty = sty; L.brackt = brackt;
        L.fx = mod ? fxw + stx * L.gtest : fxw;
        L.fy = mod ? fyw + sty * L.gtest : fyw;
        L.gx = mod ? gxw + L.gtest : gxw;
        L.gy = mod ? gyw + L.gtest : gyw;
    }
    if (L.brackt) {
        if (pw_abs(L.sty - L.stx) >= p66 * L.width1) st = L.stx + p5 * (L.sty - L.stx);
        L.width1 = L.width;
        L.width = pw_abs(L.sty - L.stx);
    }
    if (L.brackt) {
        L.stmin = pw_min(L.stx, L.sty);
        L.stmax = pw_max(L.stx, L.sty);
    } else {
        L.stmin = st + xtrapl * (st - L.stx);
        L.stmax = st + xtrapu * (st - L.stx);
    }
    st = pw_max(st, stpmin);
    st = pw_min(st, stpmax);
    if ((L.brackt && (st <= L.stmin || st >= L.stmax)) || (L.brackt && L.stmax - L.stmin <= xtol * L.stmax))
        st = L.stx;
    return LsOut{st, 1};
}
// Reverse-communication form (Lbfgsb::step, which the lockstep tests drive): one call of the search, out of line, its
// state in team memory between calls.
PW_NOINLINE PW_HD inline LsOut lb_dcsrch(LsState* lsp, int ls_task, double fv, double gv, double st, double ftol, double gtol,
                                     double xtol, double stpmin, double stpmax) {
    PW_ASSUME_LDS(lsp);
    LsState L;
    if (ls_task == 0) {
        L = LsState{};
    } else {
        L = *lsp;
    }
    const LsOut o = lb_dcsrch_core(L, ls_task, fv, gv, st, ftol, gtol, xtol, stpmin, stpmax);
    if (!(ls_task == 0 && o.task == 4)) *lsp = L;
    return o;
}


template <int N>
struct Lbfgsb {
    static constexpr int M = LB_M;
    static constexpr int M2 = 2 * LB_M;
    LbMem<N>* mem;
    // ---- problem ----
    double *l, *u;
    int* nbd;
    double *x, *g;
    double f;
    double factr, pgtol;
    int maxls;
    // ---- limited-memory matrices ----
    double *ws, *wy;
    double *sy, *ss, *wt;
    double *wn, *wn1d;
    double *z, *r, *d, *t, *xp;
    double* wa;
    double* acc;
    unsigned long long* prof;  // stage timers (diagnostic builds only), else null
    int *index, *iwhere, *indx2;
    // ---- scalars kept between calls ----
    int task, msg;
    bool prjctd, cnstnd, boxed, updatd;
    int nintol, itfile, iback, nskip, head, col, itail, iter, iupdat, nseg, nfgv, info, ifun,
        iword, nfree, nact, ileave, nenter;
    double theta, fold, tol, dnorm, epsmch, gd, stpmx, sbgnrm, stp, gdold, dtd;
    // line search (dcsrch): the rest of its state is LbMem::ls
    int ls_task;  // 0 START, 1 FG, 2 CONVERGENCE, 3 WARNING, 4 ERROR

    // ------------------------------------------------------------------
    PW_HD void bind(LbMem<N>* m) {
        mem = m;
        l = m->l; u = m->u; nbd = m->nbd; x = m->x; g = m->g;
        ws = m->ws; wy = m->wy; sy = m->sy; ss = m->ss; wt = m->wt; wn = m->wn; wn1d = m->wn1d;
        z = m->z; r = m->r; d = m->d; t = m->t; xp = m->xp; wa = m->wa; acc = m->acc;
        index = m->index; iwhere = m->iwhere; indx2 = m->indx2;
    }
    // The arrays start from ZERO, as SciPy's workspace does (_lbfgsb_py.py: wa = zeros(...), iwa =
    // zeros(...)) -- and the algorithm does read entries it never wrote: when an iteration ends with
    // every variable on a bound, formk is skipped and the row of WN1 that belongs to the newest
    // correction pair is never formed; the next formk "modifies the old parts" and reads it.  Team-shared
    // memory holds whatever the previous workgroup left, so without this the result of such a unit
    // depended on its predecessor on the CU (found as a 1-in-8192 run-to-run difference).
    template <class T>
    PW_HD void setup(LbMem<N>* m, const double* x0, const double* lo, const double* up, const int* nb,
                     double factr_, double pgtol_, int maxls_) {
        PW_ASSUME_LDS(m);
        {
            static_assert(sizeof(LbMem<N>) % 4 == 0, "LbMem is cleared in 32-bit words");
            PW_LDS int* w = (PW_LDS int*)m;
            for (int i = T::lane(); i < (int)(sizeof(LbMem<N>) / 4); i += T::WSIZE) w[i] = 0;
            T::wave_sync();
        }
        bind(m);
        if (T::WSIZE == 64) {
            // lane p of a wave owns the pair (a, b), a > b, with p = a (a - 1) / 2 + b (45 pairs for m = 10)
            const int p = T::lane();
            int a = 1;
            for (int t = 2; t < M; ++t) a += p >= t * (t - 1) / 2 ? 1 : 0;
            const int b = p - a * (a - 1) / 2;
            m->pair[p] = p < M * (M - 1) / 2 ? (a | b << 8) : 255;
            T::wave_sync();
        }
        for (int i = 0; i < N; ++i) {
            x[i] = x0[i];
            l[i] = lo[i];
            u[i] = up[i];
            nbd[i] = nb[i];
            g[i] = 0.0;
            m->le[i] = (nb[i] == 1 || nb[i] == 2) ? lo[i] : -PW_INF;
            m->ue[i] = (nb[i] == 2 || nb[i] == 3) ? up[i] : PW_INF;
        }
        f = 0.0;
        factr = factr_;
        pgtol = pgtol_;
        maxls = maxls_;
        task = LB_START;
        msg = 0;
        prof = nullptr;
    }

    PW_HD double& SY(int i, int j) { return sy[i + M * j]; }
    PW_HD double& SS(int i, int j) { return ss[i + M * j]; }
    PW_HD double& WT(int i, int j) { return wt[i + M * j]; }
    PW_HD double& WN(int i, int j) { return wn[i + M2 * j]; }
    PW_HD double& WN1(int i, int j) { return i == j ? wn1d[i] : wn[i + M2 * j]; }   // i >= j only
    PW_HD double* WS(int j) { return ws + j * N; }
    PW_HD double* WY(int j) { return wy + j * N; }


    // ---- team-parallel dense kernels (element-wise identical to pw_blas.hpp) -------------
    template <class T>
    PW_HD int p_dpotrf_u(int n, double* a, int lda) {
        PW_ASSUME_LDS(mem);
        for (int j = 0; j < n; ++j) {
            double* cj = a + (long)j * lda;
            double ajj = cj[j] - b_ddot(j, cj, cj);
            if (ajj <= 0.0) {
                T::wave_sync();
                if (T::lane() == 0) cj[j] = ajj;
                T::wave_sync();
                return j + 1;
            }
            ajj = pw_sqrt(ajj);
            int rem = n - j - 1;
            double inv = 1.0 / ajj;
            for (int k = T::lane(); k < rem; k += T::WSIZE) {
                double* ck = a + (long)(j + 1 + k) * lda;
                double y = b_dgemv_t_elem(j, rem, k, ck, cj, ck[j]);
                ck[j] = y * inv;
            }
            T::wave_sync();
            if (T::lane() == 0) cj[j] = ajj;
            T::wave_sync();
        }
        return 0;
    }
    // solve U x = b (no-trans), one right-hand side: level-2 TRSV order
    template <class T>
    PW_HD void p_dtrsv_un(int n, const double* a, int lda, double* x) {
        PW_ASSUME_LDS(mem);
        if (T::WSIZE == 64 && n <= 64) { lb_trsv_un_wave<T>(n, a, lda, x); return; }
        for (int i = n - 1; i >= 0; --i) {
            const double* ci = a + (long)i * lda;
            double xi = x[i] / ci[i];
            T::wave_sync();
            if (T::lane() == 0) x[i] = xi;
            double nx = -xi;
            for (int k = T::lane(); k < i; k += T::WSIZE) x[k] = pw_fma(nx, ci[k], x[k]);
            T::wave_sync();
        }
    }
    // solve U^T x = b, one right-hand side: DOT order (b_ddot), as a systolic sweep
    template <class T>
    PW_HD void p_dtrsv_ut(int n, const double* a, int lda, double* x) {
        PW_ASSUME_LDS(mem);
        if (T::WSIZE == 64 && n <= 64) { lb_trsv_ut_wave<T>(n, a, lda, x); return; }
        for (int i = T::lane(); i < n; i += T::WSIZE) acc[i] = 0.0;
        T::wave_sync();
        for (int s = 0; s < n; ++s) {
            double xs = x[s];
            if (s > 0) xs = xs - acc[s];
            xs = xs / a[s + (long)s * lda];
            T::wave_sync();
            if (T::lane() == 0) x[s] = xs;
            for (int i = s + 1 + T::lane(); i < n; i += T::WSIZE) {
                const double* ci = a + (long)i * lda;
                if (i >= 16 && s < 16) {
                    // rows 16.. take their first 16 terms through the SIMD ddot kernel order
                    if (s == 15) {
                        double sl[4];
                        for (int l = 0; l < 4; ++l) {
                            double x12 = (12 + l == 15) ? xs : x[12 + l];
                            double a0 = ci[l] * x[l];
                            double a1 = ci[4 + l] * x[4 + l];
                            double a2 = ci[8 + l] * x[8 + l];
                            double a3 = ci[12 + l] * x12;
                            sl[l] = ((a0 + a1) + a2) + a3;
                        }
                        acc[i] = (sl[0] + sl[2]) + (sl[1] + sl[3]);
                    }
                } else {
                    acc[i] = pw_fma(xs, ci[s], acc[i]);
                }
            }
            T::wave_sync();
        }
    }
    // dinv: 1 / a(i, i) where the caller has them tabulated (the quotients trsm forms itself otherwise), else null
    template <class T>
    PW_HD int p_dtrtrs_u(bool trans, int n, int nrhs, const double* a, int lda, double* b, int ldb,
                         const double* dinv = nullptr) {
        PW_ASSUME_LDS(mem);
        for (int i = 0; i < n; ++i)
            if (a[i + (long)i * lda] == 0.0) return i + 1;
        if (nrhs == 1) {
            if (trans) p_dtrsv_ut<T>(n, a, lda, b);
            else p_dtrsv_un<T>(n, a, lda, b);
        } else {
            for (int c = T::lane(); c < nrhs; c += T::WSIZE) b_dtrsm_ut_col(n, a, lda, b + (long)c * ldb, dinv);
            T::wave_sync();
        }
        return 0;
    }

    // ---- the three Cholesky factorisations: WT (which = 1), the first and the second diagonal block of WN (2, 3) ----
    // with the tables the solves use.  One wave: the matrix in registers (lb_potrf_wave); else, and when a pivot is
    // not positive, the team-memory routine.
    template <class T>
    PW_HD int factor(int which) {
        PW_ASSUME_LDS(mem);
        double* a = which == 1 ? wt : (which == 2 ? wn : &WN(col, col));
        const int lda = which == 1 ? M : M2;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_LB_OLD_SOLVES) && !defined(PW_LB_OLD_POTRF)
        if (T::WSIZE == 64) {
            double* rt = which == 1 ? mem->rwt : (which == 2 ? mem->rwn : mem->rwn + col);
            double* it = which == 2 ? mem->iwn : (double*)nullptr;
            if (lb_potrf_wave<LbWave, N>(mem, a, lda, col, rt, it) == 0) return 0;
        }
#endif
        const int inf = p_dpotrf_u<T>(col, a, lda);
        if (inf == 0) tables<T>(which);
        return inf;
    }

    // ---- tables of divisors and their reciprocals (device teams; LbMem) -----
    // which: 0 the diagonal of SY and its square root (after matupd), 1 the diagonal of WT (after formt),
    // 2 / 3 the first / second half of the diagonal of WN (after each of formk's factorisations)
    template <class T>
    PW_HD void tables(int which) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_LB_OLD_SOLVES)
        PW_ASSUME_LDS(mem);
        if (T::WSIZE == 64) {
            const int i = T::lane();
            if (i < col) {
                if (which == 0) {
                    const double sv = SY(i, i), q = pw_sqrt(sv);
                    mem->dsy[i] = sv; mem->rsy[i] = pw_recip_hw(sv); mem->sqy[i] = q; mem->rsq[i] = pw_recip_hw(q);
                } else if (which == 1) {
                    mem->rwt[i] = pw_recip_hw(WT(i, i));
                } else if (which == 2) {
                    mem->rwn[i] = pw_recip_hw(WN(i, i));
                    mem->iwn[i] = 1.0 / WN(i, i);
                } else {
                    mem->rwn[col + i] = pw_recip_hw(WN(col + i, col + i));
                }
            }
            T::wave_sync();
        }
#else
        (void)which;
#endif
    }

    // ---- projgr: infinity norm of the projected gradient ------------------
    PW_HD void projgr() {
        PW_ASSUME_LDS(mem);
        sbgnrm = 0.0;
        for (int i = 0; i < N; ++i) {
            double gi = g[i];
            if (nbd[i] != 0) {
                if (gi < 0.0) {
                    if (nbd[i] >= 2) gi = pw_max(x[i] - u[i], gi);
                } else {
                    if (nbd[i] <= 2) gi = pw_min(x[i] - l[i], gi);
                }
            }
            sbgnrm = pw_max(sbgnrm, pw_abs(gi));
        }
    }

    // ---- active: project x0, classify variables ----------------------------
    PW_HD void active() {
        PW_ASSUME_LDS(mem);
        prjctd = false;
        cnstnd = false;
        boxed = true;
        for (int i = 0; i < N; ++i) {
            if (nbd[i] > 0) {
                if (nbd[i] <= 2 && x[i] <= l[i]) {
                    if (x[i] < l[i]) { prjctd = true; x[i] = l[i]; }
                } else if (nbd[i] >= 2 && x[i] >= u[i]) {
                    if (x[i] > u[i]) { prjctd = true; x[i] = u[i]; }
                }
            }
        }
        for (int i = 0; i < N; ++i) {
            if (nbd[i] != 2) boxed = false;
            if (nbd[i] == 0) {
                iwhere[i] = -1;
            } else {
                cnstnd = true;
                if (nbd[i] == 2 && u[i] - l[i] <= 0.0) iwhere[i] = 3;
                else iwhere[i] = 0;
            }
        }
    }

    // ---- one or two correction pairs: the 2 col x 2 col systems of formk and subsm in scalars ---------------------
    // The statements of the BLAS restatements (pw_blas.hpp: potf2 'U', trsv 'U' 'T' / 'N' with true divisions, trsm's
    // reciprocal of the diagonal for several right-hand sides, ddot's FMA chain from +0.0) written out for n = 1, 2, 4
    // on named scalars: every lane computes the same values, nothing goes through team memory between the first load
    // and the last store, and nothing is indexed (a local array indexed in a loop lives in scratch memory on the GPU).
    // potf2 'U' of [a b; . c]: returns 0, or the 1-based index of the first non-positive pivot (stored, as potf2 does)
    PW_HD static int potf2_1(double& a) {
        const double t = a - 0.0;
        if (t <= 0.0) { a = t; return 1; }
        a = pw_sqrt(t);
        return 0;
    }
    PW_HD static int potf2_2(double& a, double& b, double& c) {
        if (potf2_1(a) != 0) return 1;
        b = b * (1.0 / a);
        const double t = c - pw_fma(b, b, 0.0);
        if (t <= 0.0) { c = t; return 2; }
        c = pw_sqrt(t);
        return 0;
    }
    // subsm's two solves (dtrtrs 'U','T' then 'U','N' on the factor of K, the first col entries negated in between)
    template <class T>
    PW_HD int subsm_solves_small(double* wv) {
        if (col == 1) {
            const double a00 = WN(0, 0), a01 = WN(0, 1), a11 = WN(1, 1);
            if (a00 == 0.0) return 1;
            if (a11 == 0.0) return 2;
            double x0 = wv[0], x1 = wv[1];
            x0 = x0 / a00;
            x1 = (x1 - pw_fma(x0, a01, 0.0)) / a11;
            x0 = -x0;
            x1 = x1 / a11;
            x0 = pw_fma(-x1, a01, x0);
            x0 = x0 / a00;
            T::wave_sync();
            wv[0] = x0; wv[1] = x1;
            T::wave_sync();
            return 0;
        }
        const double a00 = WN(0, 0), a01 = WN(0, 1), a02 = WN(0, 2), a03 = WN(0, 3);
        const double a11 = WN(1, 1), a12 = WN(1, 2), a13 = WN(1, 3);
        const double a22 = WN(2, 2), a23 = WN(2, 3), a33 = WN(3, 3);
        if (a00 == 0.0) return 1;
        if (a11 == 0.0) return 2;
        if (a22 == 0.0) return 3;
        if (a33 == 0.0) return 4;
        double x0 = wv[0], x1 = wv[1], x2 = wv[2], x3 = wv[3];
        x0 = x0 / a00;
        x1 = (x1 - pw_fma(x0, a01, 0.0)) / a11;
        x2 = (x2 - pw_fma(x1, a12, pw_fma(x0, a02, 0.0))) / a22;
        x3 = (x3 - pw_fma(x2, a23, pw_fma(x1, a13, pw_fma(x0, a03, 0.0)))) / a33;
        x0 = -x0; x1 = -x1;
        x3 = x3 / a33;
        { const double nx = -x3; x0 = pw_fma(nx, a03, x0); x1 = pw_fma(nx, a13, x1); x2 = pw_fma(nx, a23, x2); }
        x2 = x2 / a22;
        { const double nx = -x2; x0 = pw_fma(nx, a02, x0); x1 = pw_fma(nx, a12, x1); }
        x1 = x1 / a11;
        x0 = pw_fma(-x1, a01, x0);
        x0 = x0 / a00;
        T::wave_sync();
        wv[0] = x0; wv[1] = x1; wv[2] = x2; wv[3] = x3;
        T::wave_sync();
        return 0;
    }
    // formk from the first factorisation on: potrf of the leading block, the solve of the off-diagonal block (one
    // right-hand side: TRSV's division; two: TRSM's reciprocals), the rank update of the trailing block and its potrf.
    // Returns formk's own codes (0, -1, -2).
    template <class T>
    PW_HD int formk_tail_small() {
        int ret = 0;
        if (col == 1) {
            double a00 = WN(0, 0), a01 = WN(0, 1), a11 = WN(1, 1);
            if (potf2_1(a00) != 0) {
                ret = -1;
            } else {
                if (a00 != 0.0) a01 = a01 / a00;
                a11 = a11 + pw_fma(a01, a01, 0.0);
                if (potf2_1(a11) != 0) ret = -2;
            }
            T::wave_sync();
            WN(0, 0) = a00; WN(0, 1) = a01; WN(1, 1) = a11;
            T::wave_sync();
            return ret;
        }
        double a00 = WN(0, 0), a01 = WN(0, 1), a02 = WN(0, 2), a03 = WN(0, 3);
        double a11 = WN(1, 1), a12 = WN(1, 2), a13 = WN(1, 3);
        double a22 = WN(2, 2), a23 = WN(2, 3), a33 = WN(3, 3);
        if (potf2_2(a00, a01, a11) != 0) {
            ret = -1;
        } else {
            if (a00 != 0.0 && a11 != 0.0) {
                const double i0 = 1.0 / a00, i1 = 1.0 / a11;
                a02 = a02 * i0; a12 = pw_fma(-a02, a01, a12); a12 = a12 * i1;
                a03 = a03 * i0; a13 = pw_fma(-a03, a01, a13); a13 = a13 * i1;
            }
            a22 = a22 + pw_fma(a12, a12, pw_fma(a02, a02, 0.0));
            a23 = a23 + pw_fma(a13, a12, pw_fma(a03, a02, 0.0));
            a33 = a33 + pw_fma(a13, a13, pw_fma(a03, a03, 0.0));
            if (potf2_2(a22, a23, a33) != 0) ret = -2;
        }
        T::wave_sync();
        WN(0, 0) = a00; WN(0, 1) = a01; WN(0, 2) = a02; WN(0, 3) = a03;
        WN(1, 1) = a11; WN(1, 2) = a12; WN(1, 3) = a13;
        WN(2, 2) = a22; WN(2, 3) = a23; WN(3, 3) = a33;
        T::wave_sync();
        return ret;
    }

    // ---- bmv: product of the 2m x 2m middle matrix with a vector ----------
    // One or two correction pairs -- every call of the one-dimensional neck search of a window fit, the first
    // iterations of every chain: the product written out.  The statements of the general form below for col = 1, 2
    // (sums from +0.0 in their order, dtrsv's divisions, the same singularity test), every lane the same scalars:
    // a dozen flops, where the lane-parallel routine -- built for twenty rows -- spends 2.6 k cycles on its out-of-line
    // call, its tables and its lane exchanges (tests/tools/profile_zsearch.py).
    template <class T>
    PW_HD int bmv_small(const double* v, double* p) {
        const double w00 = WT(0, 0);
        if (w00 == 0.0) return 1;
        const double sy00 = SY(0, 0);
        const double q0 = pw_sqrt(sy00);
        if (col == 1) {
            const double v0 = v[0], v1 = v[1];
            double p1 = v1 / w00;                   // U^T x = b
            const double p0 = v0 / q0;
            p1 = p1 / w00;                          // U x = b
            const double pi0 = -p0 / q0;
            p[0] = pi0 + 0.0;
            p[1] = p1;
            T::wave_sync();
            return 0;
        }
        const double w11 = WT(1, 1), w01 = WT(0, 1);
        if (w11 == 0.0) return 2;
        const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
        const double sy10 = SY(1, 0), sy11 = SY(1, 1);
        const double q1 = pw_sqrt(sy11);
        double p2 = v2;
        double p3 = v3 + (0.0 + sy10 * v0 / sy00);
        // U^T x = b (dtrsv 'U', 'T': x[0] / u00, then the DOT accumulator from +0.0)
        p2 = p2 / w00;
        p3 = (p3 - pw_fma(p2, w01, 0.0)) / w11;
        const double p0 = v0 / q0, p1 = v1 / q1;
        // U x = b (dtrsv 'U', 'N': back substitution, AXPY order)
        p3 = p3 / w11;
        p2 = pw_fma(-p3, w01, p2);
        p2 = p2 / w00;
        const double pi0 = -p0 / q0, pi1 = -p1 / q1;
        p[0] = pi0 + (0.0 + sy10 * p3 / sy00);
        p[1] = pi1 + 0.0;
        p[2] = p2;
        p[3] = p3;
        T::wave_sync();
        return 0;
    }
    template <class T>
    PW_HD int bmv(const double* v, double* p) {
        PW_ASSUME_LDS(mem);
        if (col == 0) return 0;
        LB_F0(fb);
#ifndef PW_LB_NO_SMALL
        if (col <= 2) {
            const int inf_s = bmv_small<T>(v, p);
            LB_F1(2, fb);
            return inf_s;
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_LB_OLD_SOLVES)
        if (T::WSIZE == 64) {
            const int inf_w = lb_bmv_wave<LbWave, N>(mem, col, v, p);
            LB_F1(2, fb);
            return inf_w;
        }
#endif
        LB_F0(fa);
        for (int i = T::lane(); i < col; i += T::WSIZE) {
            if (i == 0) {
                p[col] = v[col];
            } else {
                double sum = 0.0;
                for (int k = 0; k < i; ++k) sum = sum + SY(i, k) * v[k] / SY(k, k);
                p[col + i] = v[col + i] + sum;
            }
        }
        T::wave_sync();
        LB_F1(3, fa);
        LB_F0(fu);
        int inf = p_dtrtrs_u<T>(true, col, 1, wt, M, p + col, col);
        LB_F1(4, fu);
        if (inf != 0) return inf;
        for (int i = T::lane(); i < col; i += T::WSIZE) p[i] = v[i] / pw_sqrt(SY(i, i));
        LB_F0(fn);
        inf = p_dtrtrs_u<T>(false, col, 1, wt, M, p + col, col);
        LB_F1(5, fn);
        if (inf != 0) return inf;
        LB_F0(fe);
        for (int i = T::lane(); i < col; i += T::WSIZE) {
            double pi = -p[i] / pw_sqrt(SY(i, i));
            double sum = 0.0;
            // (forming these quotients on one lane each and adding them up afterwards was measured:
            // the optimiser launch got 25 % SLOWER -- the line search, which does not even call this,
            // went from 190 to 460 us per unit; the register allocation of the inlined step is that
            // fragile.  The same transformation in subsm, below, is a gain.)
            for (int k = i + 1; k < col; ++k) sum = sum + SY(k, i) * p[col + k] / SY(i, i);
            p[i] = pi + sum;
        }
        T::wave_sync();
        LB_F1(6, fe);
        LB_F1(2, fb);
        return 0;
    }

    // ---- hpsolb: heap of breakpoints --------------------------------------
    PW_HD static void hpsolb(int n, double* tt, int* iorder, int iheap) {
        // 1-based semantics on 0-based storage: element k is tt[k-1]
        if (iheap == 0) {
            for (int k = 2; k <= n; ++k) {
                double ddum = tt[k - 1];
                int indxin = iorder[k - 1];
                int i = k;
                while (i > 1) {
                    int j = i / 2;
                    if (ddum < tt[j - 1]) {
                        tt[i - 1] = tt[j - 1];
                        iorder[i - 1] = iorder[j - 1];
                        i = j;
                    } else break;
                }
                tt[i - 1] = ddum;
                iorder[i - 1] = indxin;
            }
        }
        if (n > 1) {
            int i = 1;
            double out = tt[0];
            int indxou = iorder[0];
            double ddum = tt[n - 1];
            int indxin = iorder[n - 1];
            for (;;) {
                int j = i + i;
                if (j <= n - 1) {
                    if (tt[j] < tt[j - 1]) j = j + 1;
                    if (tt[j - 1] < ddum) {
                        tt[i - 1] = tt[j - 1];
                        iorder[i - 1] = iorder[j - 1];
                        i = j;
                        continue;
                    }
                }
                break;
            }
            tt[i - 1] = ddum;
            iorder[i - 1] = indxin;
            tt[n - 1] = out;
            iorder[n - 1] = indxou;
        }
    }

    // ---- cauchy: generalized Cauchy point ----------------------------------
    // workspace: p = wa[0..2m), c = wa[2m..4m), wbp = wa[4m..6m), v = wa[6m..8m)
    template <class T>
    PW_HD int cauchy() {
        PW_ASSUME_LDS(mem);
        double* p = wa;
        double* c = wa + 2 * M;
        double* wbp = wa + 4 * M;
        double* v = wa + 6 * M;
        double* xcp = z;
        int* iorder = indx2;  // scratch for breakpoint order (rewritten by freev afterwards)
        double* tt = t;
        if (sbgnrm <= 0.0) {
            b_dcopy(N, x, xcp);
            return 0;
        }
        bool bnded = true;
        int nfree_l = N + 1;   // 1-based position, as in the statement
        int nbreak = 0;
        int ibkmin = 0;
        double bkmin = 0.0;
        int col2 = 2 * col;
        double f1 = 0.0;
        double tl = 0.0, tu = 0.0;
        LB_F0(fc);
        // The classification of the variables is the same scalar code in every lane; what is spread over the lanes is
        // p = W^T d (lane j owns p[j] and p[col + j]; the sum over the variables i stays sequential).  A wave keeps its
        // two sums in registers and its rows of WY / WS, like the N-vectors, are read once, up front -- as LDS
        // read-modify-writes every variable cost three dependent round trips.
        constexpr bool WAVE = T::WSIZE == 64;
        const int lj = T::lane();
        const bool lon = WAVE && lj < col;
        double wyl[N], wsl[N], pa = 0.0, pb = 0.0;
        {
            const int pj = (head + (lon ? lj : 0)) % M;
            for (int i = 0; i < N; ++i) { wyl[i] = lon ? WY(pj)[i] : 0.0; wsl[i] = lon ? WS(pj)[i] : 0.0; }
        }
        double gl[N], xl[N], ll[N], ul[N], dl[N];
        int nbl[N], iwl[N];
        for (int i = 0; i < N; ++i) { gl[i] = g[i]; xl[i] = x[i]; ll[i] = l[i]; ul[i] = u[i]; nbl[i] = nbd[i]; iwl[i] = iwhere[i]; }
        if (!WAVE) {
            for (int i = T::lane(); i < col2; i += T::WSIZE) p[i] = 0.0;
            T::wave_sync();
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double neggi = -gl[i];
            if (iwl[i] != 3 && iwl[i] != -1) {
                if (nbl[i] <= 2) tl = xl[i] - ll[i];
                if (nbl[i] >= 2) tu = ul[i] - xl[i];
                bool xlower = nbl[i] <= 2 && tl <= 0.0;
                bool xupper = nbl[i] >= 2 && tu <= 0.0;
                iwl[i] = 0;
                if (xlower) {
                    if (neggi <= 0.0) iwl[i] = 1;
                } else if (xupper) {
                    if (neggi >= 0.0) iwl[i] = 2;
                } else {
                    if (pw_abs(neggi) <= 0.0) iwl[i] = -3;
                }
            }
            int pointr = head;
            if (iwl[i] != 0 && iwl[i] != -1) {
                dl[i] = 0.0;
            } else {
                dl[i] = neggi;
                f1 = f1 - neggi * neggi;
                if (WAVE) {
                    pa = pa + wyl[i] * neggi;
                    pb = pb + wsl[i] * neggi;
                } else {
                    for (int j = T::lane(); j < col; j += T::WSIZE) {
                        int pj = (pointr + j) % M;
                        p[j] = p[j] + WY(pj)[i] * neggi;
                        p[col + j] = p[col + j] + WS(pj)[i] * neggi;
                    }
                }
                if (nbl[i] <= 2 && nbl[i] != 0 && neggi < 0.0) {
                    nbreak += 1;
                    iorder[nbreak - 1] = i;
                    const double tb = tl / (-neggi);
                    tt[nbreak - 1] = tb;
                    if (nbreak == 1 || tb < bkmin) {
                        bkmin = tb;
                        ibkmin = nbreak;
                    }
                } else if (nbl[i] >= 2 && neggi > 0.0) {
                    nbreak += 1;
                    iorder[nbreak - 1] = i;
                    const double tb = tu / neggi;
                    tt[nbreak - 1] = tb;
                    if (nbreak == 1 || tb < bkmin) {
                        bkmin = tb;
                        ibkmin = nbreak;
                    }
                } else {
                    nfree_l -= 1;
                    iorder[nfree_l - 1] = i;
                    if (pw_abs(neggi) > 0.0) bnded = false;
                }
            }
        }
        for (int i = 0; i < N; ++i) { iwhere[i] = iwl[i]; d[i] = dl[i]; }
        if (WAVE) {
            if (lon) {
                p[lj] = pa;
                p[col + lj] = theta != 1.0 ? theta * pb : pb;
            }
            T::wave_sync();
        } else {
            T::wave_sync();
            if (theta != 1.0)
                for (int j = T::lane(); j < col; j += T::WSIZE) p[col + j] = theta * p[col + j];
        }
        b_dcopy(N, x, xcp);
        if (nbreak == 0 && nfree_l == N + 1) { T::wave_sync(); return 0; }
        for (int j = T::lane(); j < col2; j += T::WSIZE) c[j] = 0.0;
        T::wave_sync();
        double f2 = -theta * f1;
        double f2_org = f2;
        LB_F1(7, fc);
        if (col > 0) {
            int inf = bmv<T>(p, v);
            if (inf != 0) return inf;
            LB_F0(fd);
            f2 = f2 - b_ddot(col2, v, p);
            LB_F1(15, fd);
        }
        LB_F0(fz);
        double dtm = -f1 / f2;
        double tsum = 0.0;
        nseg = 1;
        bool skip_to_999 = false;
        if (nbreak != 0) {
            int nleft = nbreak;
            int it = 1;
            double tj = 0.0;
            for (;;) {
                double tj0 = tj;
                int ibp;
                if (it == 1) {
                    tj = bkmin;
                    ibp = iorder[ibkmin - 1];
                } else {
                    if (it == 2) {
                        if (ibkmin != nbreak) {
                            tt[ibkmin - 1] = tt[nbreak - 1];
                            iorder[ibkmin - 1] = iorder[nbreak - 1];
                        }
                    }
                    hpsolb(nleft, tt, iorder, it - 2);
                    tj = tt[nleft - 1];
                    ibp = iorder[nleft - 1];
                }
                double dt = tj - tj0;
                if (dtm < dt) break;  // minimizer within this interval -> 888
                tsum = tsum + dt;
                nleft -= 1;
                it += 1;
                double dibp = d[ibp];
                d[ibp] = 0.0;
                double zibp;
                if (dibp > 0.0) {
                    zibp = u[ibp] - x[ibp];
                    xcp[ibp] = u[ibp];
                    iwhere[ibp] = 2;
                } else {
                    zibp = l[ibp] - x[ibp];
                    xcp[ibp] = l[ibp];
                    iwhere[ibp] = 1;
                }
                if (nleft == 0 && nbreak == N) {
                    dtm = dt;
                    skip_to_999 = true;
                    break;
                }
                nseg += 1;
                double dibp2 = dibp * dibp;
                f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
                f2 = f2 - theta * dibp2;
                if (col > 0) {
                    for (int j = T::lane(); j < col2; j += T::WSIZE) c[j] = pw_fma(dt, p[j], c[j]);
                    for (int j = T::lane(); j < col; j += T::WSIZE) {
                        int pj = (head + j) % M;
                        wbp[j] = WY(pj)[ibp];
                        wbp[col + j] = theta * WS(pj)[ibp];
                    }
                    T::wave_sync();
                    int inf = bmv<T>(wbp, v);
                    if (inf != 0) return inf;
                    double wmc = b_ddot(col2, c, v);
                    double wmp = b_ddot(col2, p, v);
                    double wmw = b_ddot(col2, wbp, v);
                    T::wave_sync();
                    for (int j = T::lane(); j < col2; j += T::WSIZE) p[j] = pw_fma(-dibp, wbp[j], p[j]);
                    T::wave_sync();
                    f1 = f1 + dibp * wmc;
                    f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
                }
                f2 = pw_max(epsmch * f2_org, f2);
                if (nleft > 0) {
                    dtm = -f1 / f2;
                    continue;
                } else if (bnded) {
                    f1 = 0.0;
                    f2 = 0.0;
                    dtm = 0.0;
                } else {
                    dtm = -f1 / f2;
                }
                break;
            }
        }
        if (!skip_to_999) {
            if (dtm <= 0.0) dtm = 0.0;
            tsum = tsum + dtm;
            b_daxpy(N, tsum, d, xcp);
        }
        if (col > 0) {
            T::wave_sync();
            for (int j = T::lane(); j < col2; j += T::WSIZE) c[j] = pw_fma(dtm, p[j], c[j]);
        }
        T::wave_sync();
        LB_F1(14, fz);
        return 0;
    }

    // ---- freev ---------------------------------------------------------------
    PW_HD bool freev() {
        PW_ASSUME_LDS(mem);
        nenter = 0;
        ileave = N + 1;  // 1-based
        if (iter > 0 && cnstnd) {
            for (int i = 0; i < nfree; ++i) {
                int k = index[i];
                if (iwhere[k] > 0) {
                    ileave -= 1;
                    indx2[ileave - 1] = k;
                }
            }
            for (int i = nfree; i < N; ++i) {
                int k = index[i];
                if (iwhere[k] <= 0) {
                    nenter += 1;
                    indx2[nenter - 1] = k;
                }
            }
        }
        bool wrk = (ileave < N + 1) || (nenter > 0) || updatd;
        nfree = 0;
        int iact = N + 1;
        for (int i = 0; i < N; ++i) {
            if (iwhere[i] <= 0) {
                nfree += 1;
                index[nfree - 1] = i;
            } else {
                iact -= 1;
                index[iact - 1] = i;
            }
        }
        return wrk;
    }

    // ---- formk ------------------------------------------------------------------
    template <class T>
    PW_HD int formk() {
        PW_ASSUME_LDS(mem);
        const int nsub = nfree;
        LB_F0(fk1);
        if (updatd) {
            if (iupdat > M) {
                // shift the old part of WN1 up-left by one: every target reads a cell of the
                // next column, so the whole shift is "read all, then write all" -- staged in
                // registers (two cells of each block per lane)
                // block (1,1) and (2,2): element (r, jy) <- (r+1, jy+1), r = jy..M-2
                // block (2,1): element (M+r, jy) <- (M+r+1, jy+1), r = 0..M-2
                constexpr int CELLS = (M - 1) * (M - 1);
                constexpr int PER = (CELLS + T::WSIZE - 1) / T::WSIZE;
                if (T::WSIZE > 1) {
                    double s11[PER], s22[PER], s21[PER];
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        s11[q] = s22[q] = s21[q] = 0.0;
                        int e = T::lane() + q * T::WSIZE;
                        if (e < CELLS) {
                            int jy = e / (M - 1), r = e % (M - 1);
                            if (r >= jy) { s11[q] = WN1(r + 1, jy + 1); s22[q] = WN1(M + r + 1, M + jy + 1); }
                            s21[q] = WN1(M + r + 1, jy + 1);
                        }
                    }
                    T::wave_sync();
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        int e = T::lane() + q * T::WSIZE;
                        if (e < CELLS) {
                            int jy = e / (M - 1), r = e % (M - 1);
                            if (r >= jy) { WN1(r, jy) = s11[q]; WN1(M + r, M + jy) = s22[q]; }
                            WN1(M + r, jy) = s21[q];
                        }
                    }
                    T::wave_sync();
                } else {
                    // one-thread team: in place, in an order that never overwrites a pending source
                    // (targets ascend column by column, sources lie one column to the right)
                    for (int jy = 0; jy < M - 1; ++jy) {
                        for (int r = jy; r < M - 1; ++r) {
                            WN1(r, jy) = WN1(r + 1, jy + 1);
                            WN1(M + r, M + jy) = WN1(M + r + 1, M + jy + 1);
                        }
                        for (int r = 0; r < M - 1; ++r) WN1(M + r, jy) = WN1(M + r + 1, jy + 1);
                    }
                }
            }
            int ipntr = head + col - 1;
            if (ipntr >= M) ipntr -= M;
            int iy = col - 1;
            int is = M + col - 1;
            for (int jy = T::lane(); jy < col; jy += T::WSIZE) {
                int jpntr = (head + jy) % M;
                int js = M + jy;
                double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0;
                for (int k = 0; k < nsub; ++k) {
                    int k1 = index[k];
                    temp1 = temp1 + WY(ipntr)[k1] * WY(jpntr)[k1];
                }
                for (int k = nsub; k < N; ++k) {
                    int k1 = index[k];
                    temp2 = temp2 + WS(ipntr)[k1] * WS(jpntr)[k1];
                    temp3 = temp3 + WS(ipntr)[k1] * WY(jpntr)[k1];
                }
                WN1(iy, jy) = temp1;
                WN1(is, js) = temp2;
                WN1(is, jy) = temp3;
            }
            T::wave_sync();
            int jy = col - 1;
            int jpntr = head + col - 1;
            if (jpntr >= M) jpntr -= M;
            for (int i = T::lane(); i < col; i += T::WSIZE) {
                int ip = (head + i) % M;
                int is2 = M + i;
                double temp3 = 0.0;
                for (int k = 0; k < nsub; ++k) {
                    int k1 = index[k];
                    temp3 = temp3 + WS(ip)[k1] * WY(jpntr)[k1];
                }
                WN1(is2, jy) = temp3;
            }
            T::wave_sync();
        }
        int upcl = updatd ? col - 1 : col;
        // (no variable entered or left the free set -- the rule while no bound is active: every sum below is an
        // empty one, and x + 0 - 0 = x for the entries of WN1, none of which is a negative zero -- they are sums
        // that start from +0 -- so the two loops are skipped)
        if (nenter > 0 || ileave <= N) {
        // modify the old parts in blocks (1,1) and (2,2): pairs (iy, jy <= iy)
        for (int e = T::lane(); e < upcl * upcl; e += T::WSIZE) {
            int iy = e / upcl, jy = e % upcl;
            if (jy > iy) continue;
            int ipntr = (head + iy) % M, jpntr = (head + jy) % M;
            int is = M + iy, js = M + jy;
            double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
            for (int k = 0; k < nenter; ++k) {
                int k1 = indx2[k];
                temp1 = temp1 + WY(ipntr)[k1] * WY(jpntr)[k1];
                temp2 = temp2 + WS(ipntr)[k1] * WS(jpntr)[k1];
            }
            for (int k = ileave - 1; k < N; ++k) {
                int k1 = indx2[k];
                temp3 = temp3 + WY(ipntr)[k1] * WY(jpntr)[k1];
                temp4 = temp4 + WS(ipntr)[k1] * WS(jpntr)[k1];
            }
            WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
            WN1(is, js) = WN1(is, js) - temp2 + temp4;
        }
        // modify the old parts in block (2,1): all (is, jy)
        for (int e = T::lane(); e < upcl * upcl; e += T::WSIZE) {
            int ii = e / upcl, jy = e % upcl;
            int is = M + ii;
            int ipntr = (head + ii) % M, jpntr = (head + jy) % M;
            double temp1 = 0.0, temp3 = 0.0;
            for (int k = 0; k < nenter; ++k) {
                int k1 = indx2[k];
                temp1 = temp1 + WS(ipntr)[k1] * WY(jpntr)[k1];
            }
            for (int k = ileave - 1; k < N; ++k) {
                int k1 = indx2[k];
                temp3 = temp3 + WS(ipntr)[k1] * WY(jpntr)[k1];
            }
            if (is <= jy + M) WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
            else WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
        }
        }
        T::wave_sync();
        // form the upper triangle of WN from WN1: pairs (iy, jy)
        for (int e = T::lane(); e < col * col; e += T::WSIZE) {
            int iy = e / col, jy = e % col;
            int is = col + iy, is1 = M + iy;
            if (jy <= iy) {
                int js = col + jy, js1 = M + jy;
                double w = WN1(iy, jy) / theta;
                if (jy == iy) w = w + SY(iy, iy);
                WN(jy, iy) = w;
                WN(js, is) = WN1(is1, js1) * theta;
            }
            if (jy < iy) WN(jy, is) = -WN1(is1, jy);
            else WN(jy, is) = WN1(is1, jy);
        }
        T::wave_sync();
        LB_F1(8, fk1);
#ifndef PW_LB_NO_SMALL
        if (col <= 2) {
            LB_F0(fks);
            const int rs = formk_tail_small<T>();
            LB_F1(9, fks);
            return rs;
        }
#endif
        LB_F0(fk2);
        int inf = factor<T>(2);
        LB_F1(9, fk2);
        if (inf != 0) return -1;
        int col2 = 2 * col;
        LB_F0(fk3);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_LB_OLD_SOLVES) && !defined(PW_LB_OLD_POTRF)
        inf = p_dtrtrs_u<T>(true, col, col, wn, M2, &WN(0, col), M2, T::WSIZE == 64 ? mem->iwn : (const double*)nullptr);
#else
        inf = p_dtrtrs_u<T>(true, col, col, wn, M2, &WN(0, col), M2);
#endif
        LB_F1(10, fk3);
        LB_F0(fk4);
        for (int e = T::lane(); e < col * col; e += T::WSIZE) {
            int is = col + e / col, js = col + e % col;
            if (js < is) continue;
            WN(is, js) = WN(is, js) + b_ddot(col, &WN(0, is), &WN(0, js));
        }
        T::wave_sync();
        inf = factor<T>(3);
        LB_F1(11, fk4);
        (void)col2;
        if (inf != 0) return -2;
        return 0;
    }

    // ---- cmprlb -------------------------------------------------------------------
    template <class T>
    PW_HD int cmprlb() {
        PW_ASSUME_LDS(mem);
        if (!cnstnd && col > 0) {
            for (int i = 0; i < N; ++i) r[i] = -g[i];
        } else {
            for (int i = 0; i < nfree; ++i) {
                int k = index[i];
                r[i] = -theta * (z[k] - x[k]) - g[k];
            }
            T::wave_sync();
            int inf = bmv<T>(wa + 2 * M, wa);
            if (inf != 0) return -8;
            LB_F0(fr);
#if defined(__HIP_DEVICE_COMPILE__)
            if (T::WSIZE == 64) {
                // lane j forms the products of correction pair j, every lane then adds them to its copy of r in
                // the reference's order (v_readlane); r is read and written once
                const int lj = T::lane();
                const bool lon = lj < col;
                const int pj = (head + (lon ? lj : 0)) % M;
                const double a1 = lon ? wa[lj] : 0.0, a2 = lon ? theta * wa[col + lj] : 0.0;
                double t1[N], t2[N], rl[N];
                for (int i = 0; i < N; ++i) {
                    const int k = i < nfree ? index[i] : 0;
                    t1[i] = WY(pj)[k] * a1;
                    t2[i] = WS(pj)[k] * a2;
                    rl[i] = r[i];
                }
#pragma unroll
                for (int jj = 0; jj < M; ++jj) {
                    if (jj >= col) continue;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i < nfree) rl[i] = rl[i] + T::bcast_u(t1[i], jj) + T::bcast_u(t2[i], jj);
                }
                T::wave_sync();
                for (int i = 0; i < N; ++i)
                    if (i < nfree) r[i] = rl[i];
                T::wave_sync();
            } else
#endif
            {
                int pointr = head;
                for (int j = 0; j < col; ++j) {
                    double a1 = wa[j];
                    double a2 = theta * wa[col + j];
                    for (int i = 0; i < nfree; ++i) {
                        int k = index[i];
                        r[i] = r[i] + WY(pointr)[k] * a1 + WS(pointr)[k] * a2;
                    }
                    pointr = (pointr + 1) % M;
                }
            }
            LB_F1(13, fr);
        }
        return 0;
    }

    // ---- subsm -----------------------------------------------------------------------
    template <class T>
    PW_HD int subsm() {
        PW_ASSUME_LDS(mem);
        const int nsub = nfree;
        double* wv = wa;
        double* dd = r;   // direction / reduced gradient
        double* xs = z;   // on entry the Cauchy point, on exit the subspace minimiser
        if (nsub <= 0) return 0;
        T::wave_sync();
        LB_F0(fsh);
        for (int i = T::lane(); i < col; i += T::WSIZE) {
            int pi = (head + i) % M;
            double temp1 = 0.0, temp2 = 0.0;
            for (int j = 0; j < nsub; ++j) {
                int k = index[j];
                temp1 = temp1 + WY(pi)[k] * dd[j];
                temp2 = temp2 + WS(pi)[k] * dd[j];
            }
            wv[i] = temp1;
            wv[col + i] = theta * temp2;
        }
        T::wave_sync();
        LB_F1(29, fsh);
        int col2 = 2 * col;
        LB_F0(fs1);
        int inf;
#ifndef PW_LB_NO_SMALL
        if (col <= 2) {
            inf = subsm_solves_small<T>(wv);
        } else
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_LB_OLD_SOLVES)
        if (T::WSIZE == 64) {
            inf = lb_subsm_solves_wave<LbWave, N>(mem, col, wv);
        } else
#endif
        {
            inf = p_dtrtrs_u<T>(true, col2, 1, wn, M2, wv, col2);
            if (inf != 0) return inf;
            for (int i = T::lane(); i < col; i += T::WSIZE) wv[i] = -wv[i];
            T::wave_sync();
            inf = p_dtrtrs_u<T>(false, col2, 1, wn, M2, wv, col2);
        }
        LB_F1(12, fs1);
        if (inf != 0) return inf;
        LB_F0(fst);
#ifdef PW_NO_SPREAD_SUBSM
        if (false) {
#else
        if (T::WSIZE == 64 && col * nsub <= 32) {
#endif
            // the col x nsub terms by one lane each (a division apiece) into team memory (LbMem::tri: the first kind at
            // [term], the second at [32 + term]), then every lane adds them to its copy of dd in the reference's order --
            // sixty reads of addresses that are the same in every lane, issued back to back (as v_readlane moves with a
            // run-time lane number they were sixty scalar round trips) -- and scales the result by 1 / theta
            const int t = T::lane();
            const int nsu = T::uniform_i(nsub), cu = T::uniform_i(col);
            const int jy_ = nsu == 3 ? t / 3 : (nsu == 2 ? t >> 1 : t), i_ = t - jy_ * nsu;
            const double rth = 1.0 / theta;
            if (jy_ < cu) {
                const int pj = (head + jy_) % M, k = index[i_];
                mem->tri[t] = WY(pj)[k] * wv[jy_] / theta;
                mem->tri[32 + t] = WS(pj)[k] * wv[cu + jy_];
            }
            double acc_[N];
            for (int i = 0; i < N; ++i) acc_[i] = dd[i < nsu ? i : 0];
            T::wave_sync();
            // straight-line: all sixty reads first (a term that does not exist reads a slot that holds something
            // else and is replaced by -0, which added to anything leaves it as it is -- +0 would turn a -0 into +0)
            double q1[M * N], q2[M * N];
#pragma unroll
            for (int jy = 0; jy < M; ++jy)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int at = (jy * nsu + i) & 31;
                    q1[jy * N + i] = mem->tri[at];
                    q2[jy * N + i] = mem->tri[32 + at];
                }
#pragma unroll
            for (int jy = 0; jy < M; ++jy)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const bool on = jy < cu && i < nsu;
                    acc_[i] = acc_[i] + (on ? q1[jy * N + i] : -0.0) + (on ? q2[jy * N + i] : -0.0);
                }
            T::wave_sync();
            for (int i = 0; i < N; ++i)
                if (i < nsu) dd[i] = rth * acc_[i];          // (b_dscal(nsub, 1 / theta, dd))
            T::wave_sync();
        } else {
            int pointr = head;
            for (int jy = 0; jy < col; ++jy) {
                int js = col + jy;
                for (int i = 0; i < nsub; ++i) {
                    int k = index[i];
                    dd[i] = dd[i] + WY(pointr)[k] * wv[jy] / theta + WS(pointr)[k] * wv[js];
                }
                pointr = (pointr + 1) % M;
            }
            b_dscal(nsub, 1.0 / theta, dd);
        }
        // projected-search safeguard (Morales & Nocedal).  Everything is read first, then written: the free variables
        // are distinct, so this is the reference's loop -- without a round trip through team memory per statement
        iword = 0;
        b_dcopy(N, xs, xp);
        {
            int kk[N], nb[N];
            double dk[N], xk[N], lk[N], uk[N];
            for (int i = 0; i < N; ++i) {
                const bool on = i < nsub;
                kk[i] = on ? index[i] : 0;
            }
            for (int i = 0; i < N; ++i) {
                dk[i] = dd[i < nsub ? i : 0]; xk[i] = xs[kk[i]]; nb[i] = nbd[kk[i]]; lk[i] = l[kk[i]]; uk[i] = u[kk[i]];
            }
            T::wave_sync();
            for (int i = 0; i < N; ++i) {
                if (i >= nsub) continue;
                double nx;
                bool hit = false;
                if (nb[i] != 0) {
                    nx = xk[i];
                    if (nb[i] == 1) {
                        nx = pw_max(lk[i], xk[i] + dk[i]);
                        hit = nx == lk[i];
                    } else if (nb[i] == 2) {
                        nx = pw_min(uk[i], pw_max(lk[i], xk[i] + dk[i]));
                        hit = nx == lk[i] || nx == uk[i];
                    } else if (nb[i] == 3) {
                        nx = pw_min(uk[i], xk[i] + dk[i]);
                        hit = nx == uk[i];
                    }
                } else {
                    nx = xk[i] + dk[i];
                }
                if (hit) iword = 1;
                xs[kk[i]] = nx;
            }
            T::wave_sync();
        }
        LB_F1(30, fst);
        if (iword == 0) return 0;
        double dd_p = 0.0;
        for (int i = 0; i < N; ++i) dd_p = dd_p + (xs[i] - x[i]) * g[i];
        if (dd_p > 0.0) {
            b_dcopy(N, xp, xs);
            double alpha = 1.0;
            double temp1 = alpha;
            int ibd = 0;
            for (int i = 0; i < nsub; ++i) {
                int k = index[i];
                double dk = dd[i];
                if (nbd[k] != 0) {
                    if (dk < 0.0 && nbd[k] <= 2) {
                        double temp2 = l[k] - xs[k];
                        if (temp2 >= 0.0) temp1 = 0.0;
                        else if (dk * alpha < temp2) temp1 = temp2 / dk;
                    } else if (dk > 0.0 && nbd[k] >= 2) {
                        double temp2 = u[k] - xs[k];
                        if (temp2 <= 0.0) temp1 = 0.0;
                        else if (dk * alpha > temp2) temp1 = temp2 / dk;
                    }
                    if (temp1 < alpha) {
                        alpha = temp1;
                        ibd = i;
                    }
                }
            }
            if (alpha < 1.0) {
                double dk = dd[ibd];
                int k = index[ibd];
                if (dk > 0.0) {
                    xs[k] = u[k];
                    dd[ibd] = 0.0;
                } else if (dk < 0.0) {
                    xs[k] = l[k];
                    dd[ibd] = 0.0;
                }
            }
            for (int i = 0; i < nsub; ++i) {
                int k = index[i];
                xs[k] = xs[k] + alpha * dd[i];
            }
        }
        return 0;
    }

    // ---- dcstep / dcsrch: lb_dcstep, lb_dcsrch (out of line, above) --------------------
    // ---- lnsrlb --------------------------------------------------------------------------
    // returns true if a new (f,g) evaluation is requested, false when the line
    // search finished (task NEW_X) or failed (info != 0)
    template <class T>
    PW_HD bool lnsrlb(bool reentry) {
        PW_ASSUME_LDS(mem);
        const double big = 1.0e10, ftol = 1.0e-3, gtol = 0.9, xtol = 0.1;
        if (!reentry) {
            dnorm = b_dnrm2(N, d);
            dtd = dnorm * dnorm;
            stpmx = big;
            if (cnstnd) {
                if (iter == 0) {
                    stpmx = 1.0;
                } else {
                    for (int i = 0; i < N; ++i) {
                        double a1 = d[i];
                        if (nbd[i] != 0) {
                            if (a1 < 0.0 && nbd[i] <= 2) {
                                double a2 = l[i] - x[i];
                                if (a2 >= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx < a2) stpmx = a2 / a1;
                            } else if (a1 > 0.0 && nbd[i] >= 2) {
                                double a2 = u[i] - x[i];
                                if (a2 <= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx > a2) stpmx = a2 / a1;
                            }
                        }
                    }
                }
            }
            if (iter == 0 && !boxed) stp = pw_min(1.0 / dnorm, stpmx);
            else stp = 1.0;
            b_dcopy(N, x, t);
            b_dcopy(N, g, r);
            fold = f;
            ifun = 0;
            iback = 0;
            ls_task = 0;
        }
        gd = b_ddot(N, g, d);
        if (ifun == 0) {
            gdold = gd;
            if (gd >= 0.0) {
                info = -4;
                return false;
            }
        }
        LB_F0(fl);
        {
            const LsOut o = lb_dcsrch(&mem->ls, ls_task, f, gd, stp, ftol, gtol, xtol, 0.0, stpmx);
            stp = o.st;
            ls_task = o.task;
        }
        LB_F1(23, fl);
        if (ls_task != 2 && ls_task != 3) {
            task = LB_FG;
            msg = LBM_FG_LNSRCH;
            ifun += 1;
            nfgv += 1;
            iback = ifun - 1;
            if (stp == 1.0) {
                b_dcopy(N, z, x);
            } else {
                // "take step and prevent rounding error beyond bound" (SciPy's lnsrlb since its fix of iterates that
                // left the box by an ulp): a step that ends ON a bound -- stp == stpmx -- is stp * d + t only up to
                // rounding, and the iterate is put back on the bound.  Found by probing the reference with random
                // molecules: one in 574 took such a step (tests/golden/bound_step.npz).
                // (max with -inf / min with +inf where a side is open: LbMem::le / ue, set up once)
                for (int i = 0; i < N; ++i) x[i] = pw_min(pw_max(stp * d[i] + t[i], mem->le[i]), mem->ue[i]);
            }
            return true;
        }
        task = LB_NEW_X;
        msg = 0;
        return false;
    }

    // ---- matupd ---------------------------------------------------------------------------
    template <class T>
    PW_HD void matupd(double rr, double dr) {
        PW_ASSUME_LDS(mem);
        if (iupdat <= M) {
            col = iupdat;
            itail = (head + iupdat - 1) % M;
        } else {
            itail = (itail + 1) % M;
            head = (head + 1) % M;
        }
        b_dcopy(N, d, WS(itail));
        b_dcopy(N, r, WY(itail));
        theta = rr / dr;
        T::wave_sync();
        if (iupdat > M) {
            // move the old information up-left by one (read everything, then write)
            constexpr int PERM = ((M - 1) * (M - 1) + T::WSIZE - 1) / T::WSIZE;     // cells per lane
            double keep_ss[PERM], keep_sy[PERM];
            int cnt = (col - 1) * (col - 1);
#pragma unroll
            for (int q = 0; q < PERM; ++q) {
                keep_ss[q] = keep_sy[q] = 0.0;
                int e = T::lane() + q * T::WSIZE;
                if (e < cnt && T::WSIZE > 1) {
                    int j = e / (col - 1), i = e % (col - 1);
                    keep_ss[q] = (i <= j) ? SS(i + 1, j + 1) : 0.0;
                    keep_sy[q] = (i >= j) ? SY(i + 1, j + 1) : 0.0;
                }
            }
            if (T::WSIZE > 1) {
                T::wave_sync();
#pragma unroll
                for (int q = 0; q < PERM; ++q) {
                    int e = T::lane() + q * T::WSIZE;
                    if (e < cnt) {
                        int j = e / (col - 1), i = e % (col - 1);
                        if (i <= j) SS(i, j) = keep_ss[q];
                        if (i >= j) SY(i, j) = keep_sy[q];
                    }
                }
            } else {
                for (int j = 0; j < col - 1; ++j) {
                    b_dcopy(j + 1, &SS(1, j + 1), &SS(0, j));
                    b_dcopy(col - (j + 1), &SY(j + 1, j + 1), &SY(j, j));
                }
            }
            T::wave_sync();
        }
        for (int j = T::lane(); j < col - 1; j += T::WSIZE) {
            int pj = (head + j) % M;
            SY(col - 1, j) = b_ddot(N, d, WY(pj));
            SS(j, col - 1) = b_ddot(N, WS(pj), d);
        }
        if (T::lane() == 0) {
            if (stp == 1.0) SS(col - 1, col - 1) = dtd;
            else SS(col - 1, col - 1) = stp * stp * dtd;
            SY(col - 1, col - 1) = dr;
        }
        T::wave_sync();
        tables<T>(0);
    }

    // ---- formt -------------------------------------------------------------------------------
    template <class T>
    PW_HD int formt() {
        PW_ASSUME_LDS(mem);
#ifndef PW_LB_NO_SMALL
        if (col <= 2) {
            // (one or two correction pairs: the statements below and potf2 in scalars, see bmv_small; the reciprocal
            // table of WT's diagonal is the lane-parallel product's, which does not run at this size)
            double w00 = theta * SS(0, 0), w01 = 0.0, w11 = 0.0;
            int inf;
            if (col == 1) {
                inf = potf2_1(w00);
            } else {
                w01 = theta * SS(0, 1);
                const double ddum = 0.0 + SY(1, 0) * SY(1, 0) / SY(0, 0);
                w11 = ddum + theta * SS(1, 1);
                inf = potf2_2(w00, w01, w11);
            }
            T::wave_sync();
            WT(0, 0) = w00;
            if (col == 2) { WT(0, 1) = w01; WT(1, 1) = w11; }
            T::wave_sync();
            return inf != 0 ? -3 : 0;
        }
#endif
        for (int e = T::lane(); e < col * col; e += T::WSIZE) {
            int i = e / col, j = e % col;
            if (j < i) continue;
            if (i == 0) {
                WT(0, j) = theta * SS(0, j);
            } else {
                double ddum = 0.0;
                for (int k = 0; k < i; ++k) ddum = ddum + SY(i, k) * SY(j, k) / SY(k, k);
                WT(i, j) = ddum + theta * SS(i, j);
            }
        }
        T::wave_sync();
        int inf = factor<T>(1);
        if (inf != 0) return -3;
        return 0;
    }

    PW_HD void refresh() {
        info = 0;
        col = 0;
        head = 0;
        theta = 1.0;
        iupdat = 0;
        updatd = false;
    }

    // ---- the line search of one iteration in one piece (direct form) ---------------------------------
    // lnsrlb + dcsrch + the caller's function-and-gradient evaluations as ONE loop: the iterate, the direction, the
    // start point and the thirteen doubles of the More-Thuente state stay in registers from the first trial step to the
    // last -- in the reverse-communication form every trial step went through step()'s entry dispatch, a round trip of
    // the state through team memory and a dozen dependent LDS reads (x, d, t, g, the bounds).  Same statements in the
    // same order as lnsrlb(false), lnsrlb(true) ..., so the same bits.  fg(x, f, g): the objective and its gradient at
    // x (N doubles each).  Returns true when the search ended at a new iterate (task NEW_X; x, g in team memory are the
    // new point's), false when it failed (info != 0 or maxls trial steps: x, g, f restored, as step() does).
    template <class T, class FG>
    PW_HD __attribute__((always_inline)) bool linesearch(FG& fg) {
        PW_ASSUME_LDS(mem);
        LB_F0(fh);
        const double big = 1.0e10, ftol = 1.0e-3, gtol = 0.9, xtol = 0.1;
        double dv[N], xv[N], zv[N], gv[N], tv[N], lev[N], uev[N];
        int nbv[N];
        for (int i = 0; i < N; ++i) {
            dv[i] = d[i]; xv[i] = x[i]; zv[i] = z[i]; gv[i] = g[i]; lev[i] = mem->le[i]; uev[i] = mem->ue[i]; nbv[i] = nbd[i];
        }
        dnorm = b_dnrm2(N, d);
        dtd = dnorm * dnorm;
        stpmx = big;
        if (cnstnd) {
            if (iter == 0) {
                stpmx = 1.0;
            } else {
                for (int i = 0; i < N; ++i) {
                    double a1 = dv[i];
                    if (nbv[i] != 0) {
                        if (a1 < 0.0 && nbv[i] <= 2) {
                            double a2 = l[i] - xv[i];
                            if (a2 >= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx < a2) stpmx = a2 / a1;
                        } else if (a1 > 0.0 && nbv[i] >= 2) {
                            double a2 = u[i] - xv[i];
                            if (a2 <= 0.0) stpmx = 0.0;
                            else if (a1 * stpmx > a2) stpmx = a2 / a1;
                        }
                    }
                }
            }
        }
        if (iter == 0 && !boxed) stp = pw_min(1.0 / dnorm, stpmx);
        else stp = 1.0;
        for (int i = 0; i < N; ++i) { tv[i] = xv[i]; t[i] = xv[i]; r[i] = gv[i]; }
        fold = f;
        ifun = 0;
        iback = 0;
        ls_task = 0;
        LsState L = LsState{};
        bool ok;
        LB_F1(26, fh);
        for (;;) {
            // gd = b_ddot(N, g, d): the sequential FMA chain of the BLAS kernel's tail (N < 16)
            double dot = 0.0;
            for (int i = 0; i < N; ++i) dot = pw_fma(dv[i], gv[i], dot);
            gd = dot;
            if (ifun == 0) {
                gdold = gd;
                if (gd >= 0.0) { info = -4; ok = false; break; }
            }
            LB_F0(fl);
            {
                const LsOut o = lb_dcsrch_core(L, ls_task, f, gd, stp, ftol, gtol, xtol, 0.0, stpmx);
                stp = o.st;
                ls_task = o.task;
            }
            LB_F1(23, fl);
            if (ls_task == 2 || ls_task == 3) {
                task = LB_NEW_X;
                msg = 0;
                ok = true;
                break;
            }
            task = LB_FG;
            msg = LBM_FG_LNSRCH;
            ifun += 1;
            nfgv += 1;
            iback = ifun - 1;
            if (stp == 1.0) {
                for (int i = 0; i < N; ++i) xv[i] = zv[i];
            } else {
                // "take step and prevent rounding error beyond bound" (see lnsrlb)
                for (int i = 0; i < N; ++i) xv[i] = pw_min(pw_max(stp * dv[i] + tv[i], lev[i]), uev[i]);
            }
            if (iback >= maxls) { ok = false; break; }
            LB_F0(fe);
            fg(xv, f, gv);
            LB_F1(25, fe);
        }
        if (ok) {
            for (int i = 0; i < N; ++i) { x[i] = xv[i]; g[i] = gv[i]; }
        } else {
            // (r holds the gradient at the start point: written above, nothing in between touches it)
            for (int i = 0; i < N; ++i) { x[i] = tv[i]; g[i] = r[i]; }
            f = fold;
        }
        T::wave_sync();
        return ok;
    }

    // ---- the driver in direct form: setulb/mainlb and SciPy's loop around it (_lbfgsb_py.py:427-456) -----------
    // What step() does between two returns, written as the loop it is, with the caller's evaluations as a function
    // object: no entry dispatch, no state that has to survive a return, and the line search in one piece (above).
    // SciPy's driver tests maxiter and maxfun at a new iterate only ("interruptions due to maxfun are postponed");
    // fg.nfev is the evaluation count it compares with maxfun.  On return task / msg say why it stopped (task ==
    // LB_NEW_X: one of the two limits), nit is the number of iterates.  step() stays: the lockstep tests drive
    // SciPy's own setulb call by call against it, and every routine it calls is the one called here.
    template <class T, class FG>
    PW_HD __attribute__((always_inline)) void minimize(FG& fg, int maxiter, int maxfun, int* nit_out) {
        PW_ASSUME_LDS(mem);
        int nit = 0;
        *nit_out = 0;
        // task == LB_START
        epsmch = 2.220446049250313e-16;
        col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = false;
        iback = 0; itail = 0; iword = 0; nact = 0; ileave = 0; nenter = 0;
        fold = 0.0; dnorm = 0.0; gd = 0.0; stpmx = 0.0; sbgnrm = 0.0; stp = 0.0;
        gdold = 0.0; dtd = 0.0;
        iter = 0; nfgv = 0; nseg = 0; nintol = 0; nskip = 0; nfree = N; ifun = 0;
        tol = factr * epsmch;
        info = 0;
        ls_task = 0;
        for (int i = 0; i < N; ++i) { index[i] = 0; indx2[i] = 0; }
        active();
        task = LB_FG;
        msg = LBM_FG_START;
        T::wave_sync();
        {
            double xv[N], gv[N];
            for (int i = 0; i < N; ++i) xv[i] = x[i];
            fg(xv, f, gv);
            for (int i = 0; i < N; ++i) g[i] = gv[i];
            T::wave_sync();
        }
        nfgv = 1;
        projgr();
        if (sbgnrm <= pgtol) {
            task = LB_CONVERGENCE;
            msg = LBM_CONV_PGTOL;
            return;
        }
        for (;;) {  // label 222
            iword = -1;
            bool wrk;
            if (!cnstnd && col > 0) {
                b_dcopy(N, x, z);
                wrk = updatd;
                nseg = 0;
            } else {
                LB_T0(tc);
                int inf = cauchy<T>();
                LB_T1(16, tc);
                if (inf != 0) { refresh(); continue; }
                nintol += nseg;
                wrk = freev();
                nact = N - nfree;
            }
            if (!(nfree == 0 || col == 0)) {
                if (wrk) {
                    LB_T0(tk);
                    int inf = formk<T>();
                    LB_T1(17, tk);
                    if (inf != 0) { refresh(); continue; }
                }
                LB_T0(tm);
                int inf = cmprlb<T>();
                LB_T1(18, tm);
                LB_T0(tsb);
                if (inf == 0) inf = subsm<T>();
                LB_T1(19, tsb);
                if (inf != 0) { refresh(); continue; }
            }
            for (int i = 0; i < N; ++i) d[i] = z[i] - x[i];
            T::wave_sync();
            LB_T0(tl);
            const bool newx = linesearch<T>(fg);
            LB_T1(20, tl);
            if (!newx) {
                if (col == 0) {
                    if (info == 0) {
                        info = -9;
                        nfgv -= 1;
                        ifun -= 1;
                        iback -= 1;
                    }
                    task = LB_ABNORMAL;
                    msg = 0;
                    iter += 1;
                    *nit_out = nit;
                    return;
                }
                if (info == 0) nfgv -= 1;
                refresh();
                continue;
            }
            iter += 1;
            projgr();
            // ---- what SciPy's loop does with a new iterate ----
            nit += 1;
            *nit_out = nit;
            if (nit >= maxiter || fg.nfev > maxfun) return;      // (task stays LB_NEW_X)
            // ---- 777 ----
            if (sbgnrm <= pgtol) {
                task = LB_CONVERGENCE;
                msg = LBM_CONV_PGTOL;
                return;
            }
            double ddum = pw_max(pw_max(pw_abs(fold), pw_abs(f)), 1.0);
            if ((fold - f) <= tol * ddum) {
                task = LB_CONVERGENCE;
                msg = LBM_CONV_FTOL;
                if (iback >= 10) info = -5;
                return;
            }
            for (int i = 0; i < N; ++i) r[i] = g[i] - r[i];
            double rr = b_dnrm2(N, r);
            rr = rr * rr;
            double dr;
            if (stp == 1.0) {
                dr = gd - gdold;
                ddum = -gdold;
            } else {
                dr = (gd - gdold) * stp;
                b_dscal(N, stp, d);
                ddum = -gdold * stp;
            }
            if (dr <= epsmch * ddum) {
                nskip += 1;
                updatd = false;
                continue;
            }
            updatd = true;
            iupdat += 1;
            LB_T0(tu);
            matupd<T>(rr, dr);
            LB_T1(21, tu);
            LB_T0(tf);
            int inf = formt<T>();
            LB_T1(22, tf);
            if (inf != 0) { refresh(); continue; }
        }
    }

    // ---- the driver (setulb/mainlb): call repeatedly ----------------------------------------
    // On return: task == LB_FG      -> evaluate f,g at x, store in f,g, call again
    //            task == LB_NEW_X   -> an iteration finished, call again to continue
    //            otherwise          -> finished (task/msg say why)
    template <class T>
    PW_HD void step() {
        PW_ASSUME_LDS(mem);
        int entry;  // 0 fresh, 1 after FG_START, 2 after FG_LNSRCH, 3 after NEW_X
        if (task == LB_START) {
            epsmch = 2.220446049250313e-16;
            col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = false;
            iback = 0; itail = 0; iword = 0; nact = 0; ileave = 0; nenter = 0;
            fold = 0.0; dnorm = 0.0; gd = 0.0; stpmx = 0.0; sbgnrm = 0.0; stp = 0.0;
            gdold = 0.0; dtd = 0.0;
            iter = 0; nfgv = 0; nseg = 0; nintol = 0; nskip = 0; nfree = N; ifun = 0;
            tol = factr * epsmch;
            info = 0;
            for (int i = 0; i < N; ++i) { index[i] = 0; indx2[i] = 0; }
            active();
            task = LB_FG;
            msg = LBM_FG_START;
            return;
        } else if (task == LB_FG && msg == LBM_FG_START) {
            entry = 1;
        } else if (task == LB_FG) {
            entry = 2;
        } else if (task == LB_NEW_X) {
            entry = 3;
        } else {
            return;
        }

        if (entry == 1) {
            nfgv = 1;
            projgr();
            if (sbgnrm <= pgtol) {
                task = LB_CONVERGENCE;
                msg = LBM_CONV_PGTOL;
                return;
            }
        }
        bool resume_ls = (entry == 2);
        bool resume_newx = (entry == 3);
        for (;;) {  // label 222
            if (!resume_ls && !resume_newx) {
                iword = -1;
                bool wrk;
                bool have_dir = false;
                if (!cnstnd && col > 0) {
                    b_dcopy(N, x, z);
                    wrk = updatd;
                    nseg = 0;
                } else {
                    LB_T0(tc);
                    int inf = cauchy<T>();
                    LB_T1(16, tc);
                    if (inf != 0) { refresh(); continue; }
                    nintol += nseg;
                    wrk = freev();
                    nact = N - nfree;
                }
                if (!(nfree == 0 || col == 0)) {
                    if (wrk) {
                        LB_T0(tk);
                        int inf = formk<T>();
                        LB_T1(17, tk);
                        if (inf != 0) { refresh(); continue; }
                    }
                    LB_T0(tm);
                    int inf = cmprlb<T>();
                    LB_T1(18, tm);
                    LB_T0(tsb);
                    if (inf == 0) inf = subsm<T>();
                    LB_T1(19, tsb);
                    if (inf != 0) { refresh(); continue; }
                }
                (void)have_dir;
                for (int i = 0; i < N; ++i) d[i] = z[i] - x[i];
            }
            if (!resume_newx) {
                LB_T0(tl);
                bool need_fg = lnsrlb<T>(resume_ls);
                LB_T1(20, tl);
                resume_ls = false;
                if (info != 0 || iback >= maxls) {
                    b_dcopy(N, t, x);
                    b_dcopy(N, r, g);
                    f = fold;
                    if (col == 0) {
                        if (info == 0) {
                            info = -9;
                            nfgv -= 1;
                            ifun -= 1;
                            iback -= 1;
                        }
                        task = LB_ABNORMAL;
                        msg = 0;
                        iter += 1;
                        return;
                    } else {
                        if (info == 0) nfgv -= 1;
                        refresh();
                        continue;
                    }
                } else if (need_fg) {
                    return;  // task = FG_LNSRCH
                } else {
                    iter += 1;
                    projgr();
                    return;  // task = NEW_X
                }
            }
            // ---- 777: re-entry after NEW_X ----
            resume_newx = false;
            if (sbgnrm <= pgtol) {
                task = LB_CONVERGENCE;
                msg = LBM_CONV_PGTOL;
                return;
            }
            double ddum = pw_max(pw_max(pw_abs(fold), pw_abs(f)), 1.0);
            if ((fold - f) <= tol * ddum) {
                task = LB_CONVERGENCE;
                msg = LBM_CONV_FTOL;
                if (iback >= 10) info = -5;
                return;
            }
            for (int i = 0; i < N; ++i) r[i] = g[i] - r[i];
            double rr = b_dnrm2(N, r);
            rr = rr * rr;
            double dr;
            if (stp == 1.0) {
                dr = gd - gdold;
                ddum = -gdold;
            } else {
                dr = (gd - gdold) * stp;
                b_dscal(N, stp, d);
                ddum = -gdold * stp;
            }
            if (dr <= epsmch * ddum) {
                nskip += 1;
                updatd = false;
                continue;
            }
            updatd = true;
            iupdat += 1;
            LB_T0(tu);
            matupd<T>(rr, dr);
            LB_T1(21, tu);
            LB_T0(tf);
            int inf = formt<T>();
            LB_T1(22, tf);
            if (inf != 0) { refresh(); continue; }
        }
    }
};

}  // namespace pw
