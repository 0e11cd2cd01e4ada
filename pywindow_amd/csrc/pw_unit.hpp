// pw_unit.hpp -- the per-unit analysis pipeline: everything
// Molecule.full_analysis() (reference molecular.py:156-202) computes for one
// (frame, molecule), executed by one team (pw_team.hpp): a workgroup on gfx950
// or a single host thread in tests/hostsim.
//
// Stage map (reference file utilities.py unless noted):
//   load_unit            coordinates -> LDS, structure-of-arrays, |r|^2 cached
//   stage_basic          molecular_weight :96, center_of_mass :127, max_dim :355,
//                        pore_diameter :375
//   stage_opt            opt_pore_diameter :400 (L-BFGS-B, pw_lbfgsb.hpp)
//   stage_average        find_average_diameter :1586, vector_analysis_reversed :1556
//   stage_windows        find_windows :1364 = sampling :1409-1434, vector_preanalysis
//                        :1132, vector_analysis :1100, DBSCAN :1478, window_analysis
//                        :1191 (L-BFGS-B on z, 20x20 grid + Nelder-Mead on xy)
//
// Distance primitive: the bit-exact restatement of sklearn's
// euclidean_distances:  g = fma(z,pz, fma(x,px, y*py)),
// d = sqrt(max(((-2 g) + |r|^2) + |p|^2, 0)), |v|^2 = (v0^2 + v2^2) + v1^2.  (The N x N call of max_dim has its
// own order, and another one on the BLAS's edge tile: GramEdgeRule below.)
//
// Parallel decomposition: bulk stages are "lanes over points" (each lane owns a
// sampling vector / grid point and loops over the atoms, which every lane reads
// from the same LDS address -> broadcast, conflict-free); the optimiser chains
// are "lanes over atoms" with an exact wave-level (value, index) min reduction.
// No floating-point sum is ever reduced across lanes: sums whose order matters
// (numpy pairwise sums, row-sequential sums) are reproduced in the reference's
// order.
#pragma once
#include "../../include/pywindow_amd.h"
#include "pw_lbfgsb.hpp"
#include "pw_math.hpp"
#include "pw_team.hpp"
#include <string.h>

// Optional in-kernel stage timers (diagnostic build only: -DPW_PROFILE): the 100 MHz
// constant clock accumulated per stage by lane 0 of each wave into TeamWorkspace::prof.
// (The sums are kept in team memory -- pw_prof_lds, one array per team -- and added to TeamWorkspace::prof once, when
// the team leaves its kernel: accumulated by global atomics, a timer's atomic was still in flight at the next wait on
// the vector-memory counter, and whatever stage that wait belonged to was charged its round trip -- 12 us under load,
// measured with two timers in a row; the diagnostic build ran 0.2 ms per analysis behind the product because of it.)
#if defined(PW_PROFILE) && defined(__HIPCC__)
__shared__ unsigned long long pw_prof_lds[32];
#define PW_PROF_BASE(ws) ((unsigned long long*)pw_prof_lds)
#endif
#if defined(PW_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define PW_T0(var) long long var = wall_clock64()
#if defined(PW_ZPROF) && defined(PW_LB_FINE)
// (the optimiser's sub-phase timers of the neck search take the window stages' slots: those stay silent)
#define PW_T1(ws, slot, var) do { (void)var; } while (0)
#elif defined(PW_BARRIER_PROF)
// (the barrier-wait build: a slot receives what the team's waves waited at barriers since the last timer closed)
#define PW_T1(ws, slot, var) do { (void)var; if (T::lane() == 0) { unsigned long long w_ = 0; \
        for (int k_ = 0; k_ < T::NWAVES; ++k_) w_ += atomicExch(&pw_bar_acc[k_], 0ull); \
        atomicAdd(&pw_prof_lds[slot], w_); } } while (0)
#else
#define PW_T1(ws, slot, var) do { if (T::lane() == 0) atomicAdd(&pw_prof_lds[slot], (unsigned long long)(wall_clock64() - var)); } while (0)
#endif
#else
#define PW_T0(var) do {} while (0)
#define PW_T1(ws, slot, var) do {} while (0)
#endif

// unroll factors of the three bulk loops (measured: tests/tools/variant_sweep.sh)
#define PW_PRAGMA_(x) _Pragma(#x)
#define PW_PRAGMA(x) PW_PRAGMA_(x)
#ifndef PW_UNROLL_LIST
#define PW_UNROLL_LIST 4
#endif
#ifndef PW_UNROLL_GAP
#define PW_UNROLL_GAP 2
#endif
#ifndef PW_UNROLL_RAY
#define PW_UNROLL_RAY 2
#endif
#ifndef PW_UNROLL_KNN
#define PW_UNROLL_KNN 2
#endif
// register tiles of the two bulk evaluations of the window search: path points / grid points per thread
// that share one pass over the atoms (6 and 7 fill a 256-register budget; smaller tiles for smaller budgets)
#ifndef PW_TILE_PATH
#define PW_TILE_PATH 6
#endif
#ifndef PW_TILE_GRID
#define PW_TILE_GRID 7
#endif

// PW_TEAM_STATE_IN_LDS (a translation unit's choice, before this header): the kernel keeps its UnitShared -- the
// table of where everything of the team lives -- and its copy of the parameters in LDS instead of on its stack,
// and the stage functions may assume so: their look-ups become ds_read instead of scratch loads, and the kernel
// has no stack object whose address escapes into a call.
#if defined(PW_TEAM_STATE_IN_LDS) && defined(__HIP_DEVICE_COMPILE__)
#define PW_ASSUME_TEAM_STATE(sh, prm) do { PW_ASSUME_LDS(&(sh)); PW_ASSUME_LDS(&(prm)); } while (0)
#define PW_ASSUME_TEAM_SH(sh) PW_ASSUME_LDS(&(sh))
#else
#define PW_ASSUME_TEAM_STATE(sh, prm) do {} while (0)
#define PW_ASSUME_TEAM_SH(sh) do {} while (0)
#endif

// diagnostic builds (-DPW_DCHECKS, tests/tools/build_debug_variant.sh): an impossible value stops the wave where
// rocgdb shows the line
#if defined(PW_DCHECKS) && defined(__HIP_DEVICE_COMPILE__)
#define PW_DCHECK(cond, code) do { if (!(cond)) asm volatile("s_mov_b32 m0, %0\n\ts_trap 2" :: "n"(code) : "memory"); } while (0)
#else
#define PW_DCHECK(cond, code) do {} while (0)
#endif

namespace pw {

constexpr double GOLDEN_ANGLE = 2.399963229728653;   // np.pi * (3 - np.sqrt(5))
constexpr double FOUR_PI = 12.566370614359172;       // 4 * np.pi
constexpr double FOUR_THIRDS_PI = 4.1887902047863905;  // 4 / 3 * np.pi
constexpr double TWO_PI = 6.283185307179586;
constexpr double ONE_PI = 3.141592653589793;

// Atoms are stored grouped by van-der-Waals radius (few distinct radii per
// molecule): within one group min_i(|r_i-p| - vdw) = sqrt(min_i |r_i-p|^2) - vdw
// exactly (sqrt is monotone and correctly rounded), so bulk evaluations need one
// square root per group instead of one per atom.  `perm` maps the stored
// position back to the caller's atom index (used for every reported index and for
// first-index tie-breaks), `inv` is its inverse.
constexpr int PW_KCLS = 8;
constexpr double PW_PATH_POINTS_MAX = 1048576.0;      // points of one path scan (PW_ST_PATH_TOO_LONG beyond)
struct ClassInfo {
    int k;                  // number of groups; 0 => more than PW_KCLS radii, no grouping used
    int off[PW_KCLS + 1];   // group g = stored positions [off[g], off[g+1])
    double vdw[PW_KCLS];
};
struct Frame {
    ldouble *x, *y, *z, *xx;
    const ldouble* vdw;
    const lint* perm;
    const PW_LDS ClassInfo* cls;
};

// ---- global-memory workspace of one team ---------------------------------------------------
// The per-vector arrays are sized by the sampling-vector capacity of the launch (p_cap, a multiple of 64,
// at least PW_P_MAX): the reference's count int(log10(4 pi R^2) * 250 * adjust) has no upper limit
// (utilities.py:1409, 1616), so the host derives the capacity from the `adjust` knobs and every team gets a
// slab of team_slab_bytes(p_cap) bytes that bind_team_slab() cuts up.  Only what does not fit the idle
// part of the team's LDS ends up here (for CC3 at the default knobs: nothing but partial path rounds).
struct TeamWorkspace {
    double* knn;              // 10 p_cap: k-NN rows (non-streamed mean), later partial path rounds
    double* pts;              // 3 p_cap: sampling vectors of find_windows
    double* vals;             // p_cap: per-ray exit distance / per-survivor 2*gap / max_dim item maxima
    int* surv_k;              // p_cap
    int* labels;              // p_cap
    unsigned char* flag;      // p_cap
    unsigned long long* adj;  // p_cap x p_cap/64 words, only for launches that run DBSCAN
    // window fits of a unit with more clusters than the record holds (PW_W_MAX): per cluster the chosen
    // vector (3) | diameter (1) | centre (3), and the "fitted" flag -- a cluster count has no upper limit
    // in the reference either (utilities.py:1481-1536)
    double* xw;               // 7 p_cap
    int* xw_ok;               // p_cap
    int p_cap;
    // neighbour tables of the sampling sphere (see NbTables below); null: none (the windowed search runs)
    const unsigned* nb_off;
    const unsigned short* nb_idx;
    const double* nb_unit;    // the P unit vectors of every tabulated P, 3 doubles per point at 3 * (nb_off[P] + k); null: none
    const double* nb_bound;
    // launch-wide list of the windows beyond PW_W_MAX (appended with an atomic counter)
    pw_extra_window* xwin;
    unsigned* xwin_count;
    unsigned xwin_cap;
    double leaf[256];
    double acc8[8 * 160];
    int leaf_tab[324];
    unsigned long long prof[32];
    const unsigned* rsq;       // VRSQRT14PD table (pw_math.hpp: rsqrt14_decode), for numpy's arccos
    pw_unit_debug* dbg_base;   // stage capture of pw_analysis_debug (one record per unit), else null
    long unit;                 // index of the unit this team is working on
};
PW_HD inline size_t team_slab_bytes(int p_cap) { return (size_t)p_cap * 192; }   // 181 bytes per vector used
PW_HD inline size_t team_adj_words(int p_cap) { return (size_t)p_cap * (size_t)(p_cap / 64); }
PW_HD inline int round_p_cap(long p) {
    if (p < PW_P_MAX) p = PW_P_MAX;
    return (int)((p + 127) & ~127l);
}
// capacity implied by the adjust knobs: log10(4 pi R^2) * 250 stays below 2100 for sphere radii up to
// 4400 A (a molecule that size is far beyond anything the path handles); a unit that asks for more is
// flagged (PW_ST_POINTS_OVERFLOW) and pw_analysis_batch re-runs with what it asked for
PW_HD inline int params_p_cap(double adjust_windows, double adjust_average) {
    double a = adjust_windows > adjust_average ? adjust_windows : adjust_average;
    if (!(a > 0.0)) a = 1.0;
    double want = 2100.0 * a;
    if (want > 4.0e6) want = 4.0e6;
    return round_p_cap((long)want + 1);
}
PW_HD inline void bind_team_slab(TeamWorkspace* ws, unsigned char* slab, int p_cap) {
    const size_t p = (size_t)p_cap;
    double* d = (double*)slab;
    ws->knn = d; d += 10 * p;
    ws->pts = d; d += 3 * p;
    ws->vals = d; d += p;
    ws->xw = d; d += 7 * p;
    int* i = (int*)d;
    ws->surv_k = i; i += p;
    ws->labels = i; i += p;
    ws->xw_ok = i; i += p;
    ws->flag = (unsigned char*)i;
    ws->p_cap = p_cap;
}

// ---- per-unit scalars kept in LDS ----------------------------------------------
struct UnitVars {
    double com[3];
    double mw;
    int mw_given;      // the batch brought its molecular weight (one molecule type: template_groups_build)
    double centroid[3];
    double maxd;
    int maxd_i, maxd_j;
    double pore_g;
    int pore_atom;
    double opt_c[3];
    double opt_g;
    int opt_atom;
    double shift[3];
    ClassInfo cls;
    int cls_cnt[PW_KCLS];
    double eps;
    double radius;
    int P;
    int n_surv;
    int n_clusters;
    int n_eval;
    int status;
    // cross-wave reduction scratch
    double red_v[16];
    int red_i[16];
    int md_list[64];   // team_max_dim_few: the atoms that can be an end of the maximum dimension (stored positions)
    int md_count;
    // ---- window search only: everything from here on is NOT allocated for the optimiser-chain
    // launch (UnitShared::bytes with nframes == 1) ----
    int win_first;     // marks where the window-search variables start (offsetof)
    // sampling vector chosen for each cluster (largest 2*gap, first occurrence)
    double win_vec[PW_W_MAX][3];
    // window results by cluster
    int win_ok[PW_W_MAX];
    double win_d[PW_W_MAX];
    double win_c[PW_W_MAX][3];
};

// LDS layout helper: everything a team needs, carved from one byte buffer.
struct UnitShared {
    PW_LDS UnitVars* v;
    ldouble* vdw;    // stored (grouped) order
    ldouble* mass;   // caller's order
    lint* perm;
    lint* inv;
    Frame A;        // input coordinates
    Frame S;        // shifted coordinates (COM frame, then pore-centre frame)
    Frame R[8];     // per-wave rotated coordinates (window frames)
    PW_LDS void* lb[8];    // per-wave optimiser state
    // DBSCAN bit sets (p_cap bits each): core points / round flags / cluster roots (window search only)
    PW_LDS unsigned long long* bits[3];
    size_t rot_words;  // 8-byte words in the rotated-frame region
    int nslots;        // waves that can fit a window at a time (rotated frames = optimiser states carved)
    // everything behind the shifted frame (window frames + optimiser states) is idle until the
    // windows are fitted and serves as scratch for the sampling stages
    PW_LDS unsigned char* scratch;
    size_t scratch_bytes;
    // nrot = rotated window frames, nlb = optimiser states (both 0..nwaves)
    // nframes = 1: one frame - the optimiser-chain launch works on the input frame only, the other
    // launches of the pipeline shift it in place; lean = 1: without the window-search variables and the DBSCAN
    // bit sets (chains, average diameter)
    PW_HD static size_t bytes(int nmax, int nrot, int nlb, int nframes = 2, int lean = 0, int pcap = PW_P_MAX) {
        size_t n = (size_t)((nmax + 1) & ~1);
        size_t b = lean ? offsetof(UnitVars, win_first) : sizeof(UnitVars);
        b = (b + 15) & ~(size_t)15;
        if (lean != 1) b += 3 * (size_t)(pcap / 64) * 8;
        b += n * 8 * 2;                       // vdw, mass
        b += n * 4 * 2;                       // perm, inv
        b += n * 8 * 4 * ((size_t)nframes + (size_t)nrot);  // A, S, R[w]
        b += (size_t)nlb * ((sizeof(LbMem<3>) + 15) & ~(size_t)15);
        return b;
    }
    PW_HD void carve(unsigned char* base, int nmax, int nrot, int nlb, int nframes = 2, int lean = 0,
                     int pcap = PW_P_MAX) {
        size_t n = (size_t)((nmax + 1) & ~1);
        PW_LDS unsigned char* p = (PW_LDS unsigned char*)base;
        v = (PW_LDS UnitVars*)p;
        p += ((lean ? offsetof(UnitVars, win_first) : sizeof(UnitVars)) + 15) & ~(size_t)15;
        bits[0] = bits[1] = bits[2] = nullptr;
        if (lean != 1) {
            const size_t bw = (size_t)(pcap / 64);
            bits[0] = (PW_LDS unsigned long long*)p; bits[1] = bits[0] + bw; bits[2] = bits[1] + bw;
            p += 3 * bw * 8;
        }
        ldouble* d = (ldouble*)p;
        vdw = d; d += n;
        mass = d; d += n;
        perm = (lint*)d; d += n / 2;
        inv = (lint*)d; d += n / 2;
        A.x = d; d += n; A.y = d; d += n; A.z = d; d += n; A.xx = d; d += n; A.vdw = vdw; A.perm = perm; A.cls = &v->cls;
        if (nframes > 1) { S.x = d; d += n; S.y = d; d += n; S.z = d; d += n; S.xx = d; d += n; }
        else { S.x = A.x; S.y = A.y; S.z = A.z; S.xx = A.xx; }   // one frame: a shift happens in place
        S.vdw = vdw; S.perm = perm; S.cls = &v->cls;
        for (int w = 0; w < 8; ++w) { R[w].x = R[w].y = R[w].z = R[w].xx = nullptr; lb[w] = nullptr; }
        scratch = (PW_LDS unsigned char*)d;
        for (int w = 0; w < nrot; ++w) {
            R[w].x = d; d += n; R[w].y = d; d += n; R[w].z = d; d += n; R[w].xx = d; d += n;
            R[w].vdw = vdw; R[w].perm = perm; R[w].cls = &v->cls;
        }
        rot_words = (size_t)nrot * 4 * n;
        nslots = nrot < nlb ? nrot : nlb;
        p = (PW_LDS unsigned char*)d;
        for (int w = 0; w < nlb; ++w) {
            lb[w] = (PW_LDS void*)p;
            p += (sizeof(LbMem<3>) + 15) & ~(size_t)15;
        }
        scratch_bytes = (size_t)(p - scratch);
    }
};

// bump allocator over the idle LDS region; returns generic pointers (LDS aperture) and
// nullptr when the request does not fit (callers then use the global workspace)
struct ScratchArena {
    unsigned char* cur;
    size_t left;
    PW_HD void init(const UnitShared& sh) { cur = (unsigned char*)sh.scratch; left = sh.scratch_bytes; }
    PW_HD void* take(size_t bytes) {
        bytes = (bytes + 15) & ~(size_t)15;
        if (bytes > left) return nullptr;
        void* r = cur;
        cur += bytes;
        left -= bytes;
        return r;
    }
};

PW_HD inline void team_atomic_max(PW_LDS int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax((int*)p, v);
#else
    if (v > *p) *p = v;
#endif
}
PW_HD inline void team_atomic_min(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
PW_HD inline void team_atomic_add(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;
#endif
}
PW_HD inline unsigned team_atomic_inc(unsigned* p) {      // global counter shared by every team of a launch
    if (!p) return 0xffffffffu;
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1u);
#else
    return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED);       // (host teams of several threads share it too)
#endif
}
PW_HD inline void team_atomic_or(PW_LDS unsigned long long* p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr((unsigned long long*)p, v);
#else
    *p |= v;
#endif
}

// ---- small numerics --------------------------------------------------------------
PW_HD inline double sq3(double a, double b, double c) { return (a * a + c * c) + b * b; }
// np.linalg.norm of a 3-vector: sqrt(ddot) with the BLAS FMA chain
PW_HD inline double norm3(double a, double b, double c) {
    return pw_sqrt(pw_fma(c, c, pw_fma(b, b, a * a)));
}
PW_HD inline double gap_atom(const Frame& F, int i, double px, double py, double pz, double pp) {
    double g = pw_fma(F.z[i], pz, pw_fma(F.x[i], px, F.y[i] * py));
    double d2 = pw_m2add(g, F.xx[i]) + pp;
    double d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
    return d - F.vdw[i];
}
// one thread, all atoms, exact value AND first argmin (caller's atom numbering)
PW_HD inline double point_gap(const Frame& F, int n, double px, double py, double pz, int* arg) {
    double pp = sq3(px, py, pz);
    double best = PW_INF;
    int bi = 0x7fffffff;
    for (int i = 0; i < n; ++i) {
        double v = gap_atom(F, i, px, py, pz, pp);
        int oi = F.perm[i];
        if (v < best || (v == best && oi < bi)) { best = v; bi = oi; }
    }
    if (arg) *arg = bi;
    return best;
}
// one thread, all atoms, VALUE only: per radius group the minimum squared distance,
// then one sqrt per group.  Bit-identical to point_gap's value.
PW_HD inline double point_gap_value(const Frame& F, int n, double px, double py, double pz) {
    const auto& C = *F.cls;
    if (C.k == 0) return point_gap(F, n, px, py, pz, nullptr);
    double pp = sq3(px, py, pz);
    double best = PW_INF;
    for (int g = 0; g < C.k; ++g) {
        double m2 = PW_INF;
        int i = C.off[g];
        const int hi = C.off[g + 1];
        // blocks of eight: all LDS reads of a block are issued before its arithmetic, so the
        // (broadcast) read latency is paid once per block, not once per atom
        for (; i + 8 <= hi; i += 8) {
            double ax[8], ay[8], az[8], aq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { ax[j] = F.x[i + j]; ay[j] = F.y[i + j]; az[j] = F.z[i + j]; aq[j] = F.xx[i + j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double gg = pw_fma(az[j], pz, pw_fma(ax[j], px, ay[j] * py));
                m2 = __builtin_fmin(m2, pw_m2add(gg, aq[j]));
            }
        }
        for (; i < hi; ++i) {
            double gg = pw_fma(F.z[i], pz, pw_fma(F.x[i], px, F.y[i] * py));
            m2 = __builtin_fmin(m2, pw_m2add(gg, F.xx[i]));
        }
        // |p|^2 is added to the minimum instead of to every term: rounding is monotone, so
        // min_i fl(a_i + pp) = fl(min_i a_i + pp) exactly
        m2 = m2 + pp;
        double d = pw_sqrt(m2 > 0.0 ? m2 : 0.0);
        best = __builtin_fmin(best, d - C.vdw[g]);
    }
    return best;
}
// NP points at once by ONE thread (register blocking): every atom is read from LDS once and
// used for all NP points, so the loop is bound by arithmetic instead of LDS reads.  Each value is
// bit-identical to point_gap_value of that point.
// cand / coff (optional): the atoms to go through instead of all of them -- stored positions, ascending, coff[g] = where
// group g's begin in cand (wave_path_candidates: every atom left out is PROVABLY farther from every one of the points
// than the value the caller compares with; the values of the points that matter are the same bits).
template <int NP>
PW_HD inline void points_gap_values(const Frame& F, int n, const double* px, const double* py, const double* pz,
                                    double* out, const lint* cand = nullptr, const lint* coff = nullptr) {
    const auto& C = *F.cls;
    if (C.k == 0) {
        for (int p = 0; p < NP; ++p) out[p] = point_gap(F, n, px[p], py[p], pz[p], nullptr);
        return;
    }
    double qx[NP], qy[NP], qz[NP], pp[NP], best[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        qx[p] = px[p]; qy[p] = py[p]; qz[p] = pz[p];
        pp[p] = sq3(qx[p], qy[p], qz[p]);
        best[p] = PW_INF;
    }
    for (int g = 0; g < C.k; ++g) {
        double m2[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) m2[p] = PW_INF;
        if (cand) {
            const int chi = coff[g + 1];
PW_PRAGMA(unroll PW_UNROLL_LIST)
            for (int c = coff[g]; c < chi; ++c) {
                const int i = cand[c];
                const double x = F.x[i], y = F.y[i], z = F.z[i], xx = F.xx[i];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    double gg = pw_fma(z, qz[p], pw_fma(x, qx[p], y * qy[p]));
                    m2[p] = __builtin_fmin(m2[p], pw_m2add(gg, xx));
                }
            }
        } else {
        const int hi = C.off[g + 1];
PW_PRAGMA(unroll PW_UNROLL_GAP)
        for (int i = C.off[g]; i < hi; ++i) {
            const double x = F.x[i], y = F.y[i], z = F.z[i], xx = F.xx[i];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                double gg = pw_fma(z, qz[p], pw_fma(x, qx[p], y * qy[p]));
                m2[p] = __builtin_fmin(m2[p], pw_m2add(gg, xx));
            }
        }
        }
        const double r = C.vdw[g];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // (|p|^2 added to the minimum, not to every term: rounding is monotone -- the same bits)
            const double m2p = m2[p] + pp[p];
            double d = pw_sqrt(m2p > 0.0 ? m2p : 0.0);
            best[p] = __builtin_fmin(best[p], d - r);
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) out[p] = best[p];
}

// one wave, atoms spread over lanes; value and first argmin (caller's numbering) in every lane
template <class T>
PW_HD inline double wave_gap(const Frame& F, int n, double px, double py, double pz, int* arg) {
    double pp = sq3(px, py, pz);
    double best = PW_INF;
    int bi = 0x7fffffff;
    for (int i = T::lane(); i < n; i += T::WSIZE) {
        double v = gap_atom(F, i, px, py, pz, pp);
        int oi = F.perm[i];
        if (v < best || (v == best && oi < bi)) { best = v; bi = oi; }
    }
    T::wave_argmin(best, bi);
    if (arg) *arg = bi;
    return best;
}

// one wave, atoms spread over lanes, VALUE only (the simplex and window-diameter evaluations
// never need the atom): no index bookkeeping, value-only reduction.  Same value as wave_gap.
template <class T>
PW_HD inline double wave_gap_value(const Frame& F, int n, double px, double py, double pz) {
    double pp = sq3(px, py, pz);
    double best = PW_INF;
    for (int i = T::lane(); i < n; i += T::WSIZE) best = __builtin_fmin(best, gap_atom(F, i, px, py, pz, pp));
    return T::wave_min(best);
}

// Four points at once (the f and the three forward-difference points of one gradient
// request): each row of 16 lanes takes one point and the atoms are spread over the row.
// Values only; identical to four wave_gap calls.
// Keep it inlined into its callers: an out-of-line copy that read the caller's private arrays px/py/pz with
// the lane-dependent index below returned wrong values on gfx950 / ROCm 7.0 (by-value arguments were fine,
// and no faster than inlining) -- measured in round 2 when the evaluation was moved out of the optimiser step.
template <class T>
PW_HD inline void wave_gap4(const Frame& F, int n, const double* px, const double* py,
                            const double* pz, double* out) {
    if (T::WSIZE == 64) {
        int g = T::lane() >> 4, l = T::lane() & 15;
        double qx = g == 0 ? px[0] : (g == 1 ? px[1] : (g == 2 ? px[2] : px[3]));
        double qy = g == 0 ? py[0] : (g == 1 ? py[1] : (g == 2 ? py[2] : py[3]));
        double qz = g == 0 ? pz[0] : (g == 1 ? pz[1] : (g == 2 ? pz[2] : pz[3]));
        double pp = sq3(qx, qy, qz);
        double best = PW_INF;
        const auto& C = *F.cls;
        const int kk = T::uniform_i(C.k);
        if (kk > 0) {
            // Atoms are stored by radius group (load_unit): per group the minimum of the squared-distance terms,
            // THEN |p|^2, the root and the radius -- one root per group and lane instead of one per atom, the same
            // bits (rounding is monotone; point_gap_value has the argument).  The rounds -- sixteen atoms of one
            // group each, a lane past the end repeats the group's last atom, which a minimum does not notice -- are
            // taken two at a time (eight LDS reads in flight).  (A software pipeline over single rounds, the reads of
            // round r + 1 issued before round r is reduced, was measured 45 % slower: its scalar control flow.)
            for (int c = 0; c < kk; ++c) {
                const int lo = T::uniform_i(C.off[c]), hi = T::uniform_i(C.off[c + 1]);
                double m2 = PW_INF;
                int base = lo;
                for (; base + 16 < hi; base += 32) {
                    int i0 = base + l, i1 = base + 16 + l;
                    i1 = i1 < hi ? i1 : hi - 1;
                    const double x0 = F.x[i0], y0 = F.y[i0], z0 = F.z[i0], s0 = F.xx[i0];
                    const double x1 = F.x[i1], y1 = F.y[i1], z1 = F.z[i1], s1 = F.xx[i1];
                    const double g0 = pw_fma(z0, qz, pw_fma(x0, qx, y0 * qy));
                    const double g1 = pw_fma(z1, qz, pw_fma(x1, qx, y1 * qy));
                    m2 = __builtin_fmin(m2, pw_m2add(g0, s0));
                    m2 = __builtin_fmin(m2, pw_m2add(g1, s1));
                }
                if (base < hi) {
                    int i0 = base + l;
                    i0 = i0 < hi ? i0 : hi - 1;
                    const double gg = pw_fma(F.z[i0], qz, pw_fma(F.x[i0], qx, F.y[i0] * qy));
                    m2 = __builtin_fmin(m2, pw_m2add(gg, F.xx[i0]));
                }
                m2 = m2 + pp;
                const double d = pw_sqrt(m2 > 0.0 ? m2 : 0.0);
                best = __builtin_fmin(best, d - C.vdw[c]);
            }
        } else {
#pragma unroll 4
            for (int i = l; i < n; i += 16) best = __builtin_fmin(best, gap_atom(F, i, qx, qy, qz, pp));
        }
        T::row_min4(best, out);
    } else {
        for (int q = 0; q < 4; ++q) out[q] = wave_gap<T>(F, n, px[q], py[q], pz[q], nullptr);
    }
}

// ---- the four-point objective near a reference point ----------------------------------------------------------
// An optimiser asks for its objective some ninety times per unit, at points that lie within a few tenths of an
// angstrom of each other, and min_i(|p - a_i| - r_i) is 1-Lipschitz in p for every atom: an atom whose gap at a
// reference point R exceeds the minimum there by more than 2 delta cannot hold the minimum anywhere within delta of
// R.  rebuild() evaluates every atom at R once (all lanes), lists the atoms within 2 delta + a margin of the minimum
// (the margin, 1e-6, is a million times the rounding error of a gap) and hands each lane of a row its share of the
// list; eval() then takes the minimum over the list only -- one or two atoms per lane instead of eleven -- while the
// four points stay within delta (1-norm, so also 2-norm) of R, and rebuilds around the new point when they do not.
// The value is the minimum of the same per-atom expressions, so it has the bits of wave_gap4 (every atom, one root
// per radius group: the same bits again, see there).  More than 32 candidates: no list, wave_gap4 itself.
#if defined(__HIP_DEVICE_COMPILE__)
struct NearList4 { int c0, c1, total; };
struct NearList1 { int mine, total; };
// every atom at the reference point (all 64 lanes), the atoms within `slack` of the minimum listed in `cand` (up to
// 32); a lane's share of the list for the row layout of wave_gap4: positions l and l + 16 of its row-local number l
template <class T>
PW_NOINLINE __device__ inline NearList4 near_rebuild4(Frame F, int n, PW_LDS int* cand, double px, double py, double pz,
                                                      double slack) {
    const int lane = T::lane();
    const double pp = sq3(px, py, pz);
    double best = PW_INF;
    for (int i = lane; i < n; i += 64) best = __builtin_fmin(best, gap_atom(F, i, px, py, pz, pp));
    best = T::wave_min(best);
    const double lim = best + slack;
    int total = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && gap_atom(F, i < n ? i : 0, px, py, pz, pp) <= lim;
        const unsigned long long m = T::ballot(in);
        const int at = total + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (in && at < 32) cand[at] = i;
        total += (int)__builtin_popcountll(m);
    }
    T::wave_sync();
    NearList4 r;
    r.c0 = r.c1 = -1;
    // (the margin covers the rounding error of a gap for coordinates up to some 3e4 A; beyond, no list)
    if (!(pp < 1.0e9)) total = 1 << 20;
    r.total = total;
    const int l = lane & 15;
    if (total <= 32) {
        if (l < total) r.c0 = cand[l];
        if (l + 16 < total) r.c1 = cand[l + 16];
    }
    T::wave_sync();
    return r;
}
// the same for one point and 64 lanes: lane k keeps the k-th listed atom (no list in memory: the lane finds the
// k-th set bit of the ballots itself)
template <class T>
PW_NOINLINE __device__ inline NearList1 near_rebuild1(Frame F, int n, double px, double py, double pz, double slack) {
    const int lane = T::lane();
    const double pp = sq3(px, py, pz);
    double best = PW_INF;
    for (int i = lane; i < n; i += 64) best = __builtin_fmin(best, gap_atom(F, i, px, py, pz, pp));
    best = T::wave_min(best);
    const double lim = best + slack;
    NearList1 r;
    r.mine = -1;
    int total = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && gap_atom(F, i < n ? i : 0, px, py, pz, pp) <= lim;
        const unsigned long long m = T::ballot(in);
        const int cnt = (int)__builtin_popcountll(m);
        const int want = lane - total;
        if (want >= 0 && want < cnt) {
            unsigned long long mm = m;
            for (int k = 0; k < want; ++k) mm &= mm - 1ull;
            r.mine = i0 + (int)__builtin_ctzll(mm);
        }
        total += cnt;
    }
    if (!(pp < 1.0e9)) total = 1 << 20;             // (see near_rebuild4)
    r.total = total;
    return r;
}
template <class T>
struct NearGap4 {
    static constexpr double DELTA = 0.3, MARGIN = 1e-6;
    double rx, ry, rz;      // reference point
    int c0, c1;             // this lane's candidates (stored positions), -1: none
    int state;              // 0 no reference yet, 1 list in use, 2 too many candidates around the reference

    __device__ void init() { rx = ry = rz = 0.0; c0 = c1 = -1; state = 0; }

    __device__ void rebuild(const Frame& F, int n, PW_LDS int* cand, double px, double py, double pz) {
        const NearList4 r = near_rebuild4<T>(F, n, cand, px, py, pz, 2.0 * DELTA + MARGIN);
        rx = px; ry = py; rz = pz;
        c0 = r.c0; c1 = r.c1;
        state = r.total <= 32 ? 1 : 2;
    }

    __device__ void eval(const Frame& F, int n, PW_LDS int* cand, const double* px, const double* py, const double* pz,
                         double* out) {
        const int g = T::lane() >> 4;
        const double qx = g == 0 ? px[0] : (g == 1 ? px[1] : (g == 2 ? px[2] : px[3]));
        const double qy = g == 0 ? py[0] : (g == 1 ? py[1] : (g == 2 ? py[2] : py[3]));
        const double qz = g == 0 ? pz[0] : (g == 1 ? pz[1] : (g == 2 ? pz[2] : pz[3]));
        const bool near = pw_abs(qx - rx) + pw_abs(qy - ry) + pw_abs(qz - rz) <= DELTA;
        if (state == 0 || !T::wave_all(near)) rebuild(F, n, cand, px[0], py[0], pz[0]);
        // (the three difference points lie 1e-8 from the first: within delta of a reference that was just moved there;
        // if they do not -- a huge relative step -- the list is not used)
        const bool near2 = pw_abs(qx - rx) + pw_abs(qy - ry) + pw_abs(qz - rz) <= DELTA;
        if (state != 1 || !T::wave_all(near2)) {
            wave_gap4<T>(F, n, px, py, pz, out);
            return;
        }
        const double pp = sq3(qx, qy, qz);
        const int i0 = c0 < 0 ? 0 : c0, i1 = c1 < 0 ? 0 : c1;
        const double x0 = F.x[i0], y0 = F.y[i0], z0 = F.z[i0], s0 = F.xx[i0], r0 = F.vdw[i0];
        const double x1 = F.x[i1], y1 = F.y[i1], z1 = F.z[i1], s1 = F.xx[i1], r1 = F.vdw[i1];
        const double g0 = pw_fma(z0, qz, pw_fma(x0, qx, y0 * qy));
        const double g1 = pw_fma(z1, qz, pw_fma(x1, qx, y1 * qy));
        const double d0 = pw_m2add(g0, s0) + pp, d1 = pw_m2add(g1, s1) + pp;
        const double v0 = pw_sqrt(d0 > 0.0 ? d0 : 0.0) - r0, v1 = pw_sqrt(d1 > 0.0 ? d1 : 0.0) - r1;
        double best = c0 < 0 ? PW_INF : v0;
        best = c1 < 0 ? best : __builtin_fmin(best, v1);
        T::row_min4(best, out);
    }
};
// One point at a time (the simplex search in a window's plane): up to 64 candidates, one per lane.
template <class T>
struct NearGap1 {
    static constexpr double DELTA = 0.3, MARGIN = 1e-6;
    double rx, ry, rz;
    int c;                  // this lane's candidate, -1: none
    int state;              // 0 no reference yet, 1 list in use, 2 too many candidates
    double ax, ay, az, aq, ar;      // ... and its coordinates, squared norm and radius: an evaluation reads no memory

    __device__ void init() { rx = ry = rz = 0.0; c = -1; state = 0; ax = ay = az = aq = ar = 0.0; }

    __device__ void rebuild(const Frame& F, int n, PW_LDS int* cand, double px, double py, double pz) {
        (void)cand;
        const NearList1 r = near_rebuild1<T>(F, n, px, py, pz, 2.0 * DELTA + MARGIN);
        rx = px; ry = py; rz = pz;
        c = r.total <= 64 ? r.mine : -1;
        state = r.total <= 64 ? 1 : 2;
        const int i = c < 0 ? 0 : c;
        ax = F.x[i]; ay = F.y[i]; az = F.z[i]; aq = F.xx[i]; ar = F.vdw[i];
    }

    __device__ double eval(const Frame& F, int n, PW_LDS int* cand, double px, double py, double pz) {
        const bool near = pw_abs(px - rx) + pw_abs(py - ry) + pw_abs(pz - rz) <= DELTA;
        if (state == 0 || !T::wave_all(near)) rebuild(F, n, cand, px, py, pz);
        if (state != 1) return wave_gap_value<T>(F, n, px, py, pz);
        // (gap_atom on the lane's own atom, from registers)
        const double g = pw_fma(az, pz, pw_fma(ax, px, ay * py));
        const double d2 = pw_m2add(g, aq) + sq3(px, py, pz);
        const double v = pw_sqrt(d2 > 0.0 ? d2 : 0.0) - ar;
        return T::wave_min(c < 0 ? PW_INF : v);
    }
};
#endif

// numpy's float64 add.reduce over a contiguous 1-D array: pairwise blocks of
// <= 128 with 8 accumulators, halves split at multiples of 8, and the outer
// iterator feeding 8192-element buffers sequentially.
PW_HD inline double np_leaf_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r = r + a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    int lim = n - (n % 8);
    for (; i < lim; i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + a[i];
    return res;
}
struct NpSeg { int off, len, stage; };
// enumerate the leaves of the pairwise recursion over [off, off+n) in order
PW_HD inline int np_leaves(int off, int n, int* loff, int* llen) {
    NpSeg st[24];
    int top = 0, nl = 0;
    st[top++] = NpSeg{off, n, 0};
    while (top) {
        NpSeg c = st[--top];
        if (c.len <= 128) {
            loff[nl] = c.off; llen[nl] = c.len; ++nl;
        } else {
            int n2 = c.len / 2;
            n2 -= n2 % 8;
            st[top++] = NpSeg{c.off + n2, c.len - n2, 0};
            st[top++] = NpSeg{c.off, n2, 0};
        }
    }
    return nl;
}
// combine leaf sums in recursion order
PW_HD inline double np_combine(int n, const double* leafsum) {
    NpSeg st[24];
    double vals[24];
    int top = 0, vtop = 0, li = 0;
    st[top++] = NpSeg{0, n, 0};
    while (top) {
        NpSeg& c = st[top - 1];
        if (c.len <= 128) {
            vals[vtop++] = leafsum[li++];
            --top;
        } else if (c.stage == 0) {
            int n2 = c.len / 2;
            n2 -= n2 % 8;
            c.stage = 1;
            st[top++] = NpSeg{c.off, n2, 0};
        } else if (c.stage == 1) {
            int n2 = c.len / 2;
            n2 -= n2 % 8;
            c.stage = 2;
            int o = c.off, l = c.len;
            st[top++] = NpSeg{o + n2, l - n2, 0};
        } else {
            double r = vals[--vtop];
            double l = vals[--vtop];
            vals[vtop++] = l + r;
            --top;
        }
    }
    return vals[0];
}
// serial, array-free version for n <= 8192 (one thread): depth-first over the same
// recursion, leaves summed as they are reached
PW_HD inline double np_sum_small(const double* a, int n) {
    NpSeg st[24];
    double vals[24];
    int top = 0, vtop = 0;
    st[top++] = NpSeg{0, n, 0};
    while (top) {
        NpSeg& c = st[top - 1];
        if (c.len <= 128) {
            vals[vtop++] = np_leaf_sum(a + c.off, c.len);
            --top;
        } else if (c.stage == 0) {
            int n2 = c.len / 2;
            n2 -= n2 % 8;
            c.stage = 1;
            st[top++] = NpSeg{c.off, n2, 0};
        } else if (c.stage == 1) {
            int n2 = c.len / 2;
            n2 -= n2 % 8;
            c.stage = 2;
            int o = c.off, l = c.len;
            st[top++] = NpSeg{o + n2, l - n2, 0};
        } else {
            double r = vals[--vtop];
            double l = vals[--vtop];
            vals[vtop++] = l + r;
            --top;
        }
    }
    return vals[0];
}
// serial version (one thread)
PW_HD inline double np_sum_serial(const double* a, int n) {
    double total = 0.0;
    bool first = true;
    for (int s = 0; s < n; s += 8192) {
        int len = n - s < 8192 ? n - s : 8192;
        int loff[160], llen[160];
        double ls[160];
        int nl = np_leaves(0, len, loff, llen);
        for (int i = 0; i < nl; ++i) ls[i] = np_leaf_sum(a + s + loff[i], llen[i]);
        double part = np_combine(len, ls);
        total = first ? part : total + part;
        first = false;
    }
    return total;
}
// ... and one whose state is scalars only (nothing a GPU compiler has to put into scratch memory): the leaves
// are visited left to right - each found by walking down the recursion from the root - and a finished node
// that is a right child is added to its parent's pending left sum on the spot.  Depth <= 8 for 8192 elements.
PW_HD inline double np_sum_lean(const double* a, int n) {
    double total = 0.0;
    bool first = true;
    for (int s = 0; s < n; s += 8192) {
        const int len = n - s < 8192 ? n - s : 8192;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0, l4 = 0.0, l5 = 0.0, l6 = 0.0, l7 = 0.0, part = 0.0;
        for (int e = 0; e < len;) {
            int off = 0, l = len, depth = 0;
            unsigned path = 0;                       // bit d: the walk went right at depth d
            while (l > 128) {
                int n2 = l / 2;
                n2 -= n2 % 8;
                if (e < off + n2) { l = n2; } else { off += n2; l -= n2; path |= 1u << depth; }
                ++depth;
            }
            double v = np_leaf_sum(a + s + off, l);
            int d = depth;
            while (d > 0 && ((path >> (d - 1)) & 1u)) {
                const int k = d - 1;
                const double lv = k == 0 ? l0 : k == 1 ? l1 : k == 2 ? l2 : k == 3 ? l3 : k == 4 ? l4 : k == 5 ? l5 : k == 6 ? l6 : l7;
                v = lv + v;
                --d;
            }
            if (d > 0) {
                const int k = d - 1;
                l0 = k == 0 ? v : l0; l1 = k == 1 ? v : l1; l2 = k == 2 ? v : l2; l3 = k == 3 ? v : l3;
                l4 = k == 4 ? v : l4; l5 = k == 5 ? v : l5; l6 = k == 6 ? v : l6; l7 = k == 7 ? v : l7;
            } else {
                part = v;
            }
            e = off + l;
        }
        total = first ? part : total + part;
        first = false;
    }
    return total;
}
// team version.  Leaves are at least 64 elements long (for n > 128), so every 64-element
// slot holds at most one leaf start; a thread finds it by walking down the recursion
// (no tables, no private stacks -- those would live in scratch memory on the GPU).  The
// eight strided accumulators of a leaf are independent sequential chains, so (leaf,
// accumulator) pairs are spread over the threads; one thread per leaf folds them exactly
// as numpy does and thread 0 combines the leaves in recursion order with its stack in
// team-shared memory.  Scratch: tab 2*128+80 ints, acc8 8*128 doubles, leafbuf 128+32 doubles.
PW_HD inline void np_descend(int len, int e, int* off_out, int* len_out) {
    int off = 0, l = len;
    while (l > 128) {
        int n2 = l / 2;
        n2 -= n2 % 8;
        if (e < off + n2) { l = n2; } else { off += n2; l -= n2; }
    }
    *off_out = off;
    *len_out = l;
}
// Where the leaves of ONE 8192-element chunk (length len) start: tab[slot] = start of the leaf that
// begins in the 64-element slot (-1: none), tab[128 + slot] = its length.  No barrier inside.
template <class T, class IP>
PW_HD inline __attribute__((always_inline)) void np_leaf_table_t(int len, IP tab) {
    const int nslot = (len + 63) >> 6;
    for (int sl = T::tid(); sl < nslot; sl += T::SIZE) {
        int e = sl * 64, off, l, start = -1, ln = 0;
        np_descend(len, e, &off, &l);
        if (off == e) { start = off; ln = l; }
        else {
            int nxt = off + l;
            int lim = e + 64 < len ? e + 64 : len;
            if (nxt < lim) { np_descend(len, nxt, &off, &l); start = off; ln = l; }
        }
        tab[sl] = start;
        tab[128 + sl] = ln;
    }
}
template <class T>
PW_HD inline void np_leaf_table(int len, int* tab) {
    if (PW_IS_LDS(tab)) np_leaf_table_t<T>(len, PW_AS_LDS(tab)); else np_leaf_table_t<T>(len, tab);
}
// Leaf sums of the chunk for the leaves that start in [e_lo, e_hi) -- whole leaves by construction:
// both bounds are leaf starts (or the chunk's end).  `src` holds the elements from e_lo on: element e of
// the chunk is src[e - e_lo].  leafbuf[slot of the leaf's start] receives the sum.  The table above must
// be visible (a barrier after np_leaf_table).  The eight strided accumulators of a leaf are eight
// adjacent lanes: each adds its (at most 16) elements in order, then the eight are folded as numpy folds
// them -- ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), by three lane exchanges (IEEE addition commutes, so
// every lane of the eight ends up with the same value) -- and lane 0 adds the tail.  One barrier, at the
// end.  (acc8 is only used by the one-thread host build, which has no lanes to exchange with.)
template <class T, class SP, class IP, class DP>
PW_HD inline __attribute__((always_inline)) void np_leaf_sums_t(SP src, int len, int e_lo, int e_hi, IP tab, DP acc8,
                                                                DP leafbuf) {
    (void)len;
    const int sl_lo = e_lo >> 6;
    const int sl_hi = (e_hi + 63) >> 6;          // exclusive
    const int nslot = sl_hi - sl_lo;
    for (int task = T::tid(); task < nslot * 8; task += T::SIZE) {
        const int sl = sl_lo + (task >> 3), c = task & 7;
        const int start = tab[sl];
        if (start < e_lo || start >= e_hi) continue;                  // no leaf here, or a leaf of another tile
        SP b = src + (start - e_lo);
        const int ln = tab[128 + sl];
        if (ln < 8) {
            if (c == 0) {
                double r = 0.0;
                for (int i = 0; i < ln; ++i) r = r + b[i];
                leafbuf[sl] = r;
            }
            continue;
        }
        // a leaf has at most 128 elements, i.e. at most 16 per accumulator: all loads first
        // (they are independent), then the additions in order
        const int lim = ln - (ln % 8);
        double vv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int i = c + 8 * u;
            vv[u] = b[i < lim ? i : c];
        }
        double r = vv[0];
#pragma unroll
        for (int u = 1; u < 16; ++u)
            if (c + 8 * u < lim) r = r + vv[u];
        double res;
        if (T::SIZE >= 8) {
            r = r + T::xor_d(r, 1);
            r = r + T::xor_d(r, 2);
            res = r + T::xor_d(r, 4);
        } else {
            // one thread plays the eight accumulators in turn
            acc8[c] = r;
            if (c != 7) continue;
            res = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        }
        if (T::SIZE < 8 || c == 0) {
            for (int i = lim; i < ln; ++i) res = res + b[i];
            leafbuf[sl] = res;
        }
    }
    T::sync();
}
template <class T>
PW_HD inline void np_leaf_phase(const double* src, int len, int e_lo, int e_hi, int* tab, double* acc8,
                                double* leafbuf, bool table_ready = false) {
    if (!table_ready) {
        np_leaf_table<T>(len, tab);
        T::sync();
    }
    if (PW_IS_LDS(src) && PW_IS_LDS(tab) && PW_IS_LDS(acc8) && PW_IS_LDS(leafbuf))
        np_leaf_sums_t<T>(PW_AS_LDS(src), len, e_lo, e_hi, PW_AS_LDS(tab), PW_AS_LDS(acc8), PW_AS_LDS(leafbuf));
    else
        np_leaf_sums_t<T>(src, len, e_lo, e_hi, tab, acc8, leafbuf);
}
// Combine the leaf sums of one chunk as numpy's recursion does: every inner node is (sum of its left
// part) + (sum of its right part), whatever order the nodes are visited in -- so the tree is folded
// level by level from the deepest one, one thread per node, instead of walking it depth first (that
// walk was a chain of ~400 dependent LDS round trips for 64 leaves).  A node is named by its depth and
// the left/right turns from the root; its (offset, length) follow from at most seven halvings.  The
// right part is the larger one, and it is a leaf (<= 128 elements) at depth 7 at the latest for a
// chunk of 8192.  The result is valid on thread 0.  acc8: at least 128 doubles; leafbuf: 256 doubles
// (leaf sums by 64-element slot in the first 128).
template <class T, class DP, class IP>
PW_HD inline __attribute__((always_inline)) double np_walk_phase_t(int len, IP tab, DP acc8, DP leafbuf) {
    int depth = 0;
    for (int l = len; l > 128; ++depth) { int n2 = l / 2; n2 -= n2 % 8; l -= n2; }
    // (one wave: its lanes see each other's LDS writes without a team barrier)
    if (T::wave() == 0) {
        // the tree first, top down, one halving per node: node t of depth d is entry (1 << d) + t of `tab` (free once the
        // leaf sums are in: offset | length << 16, length 0: no such node).  Naming a node by the turns from the root and
        // re-deriving its extent at every level was seven halvings per node and level -- most of this phase.
        if (T::lane() == 0) tab[1] = len << 16;
        T::wave_sync();
        for (int d = 1; d <= depth; ++d) {
            for (int t = T::lane(); t < (1 << d); t += T::WSIZE) {
                const int par = tab[((1 << d) + t) >> 1];
                const int po = par & 0xffff, pl = par >> 16;
                int e = 0;
                if (pl > 128) {
                    int n2 = pl / 2;
                    n2 -= n2 % 8;
                    e = (t & 1) ? ((po + n2) | ((pl - n2) << 16)) : (po | (n2 << 16));
                }
                tab[(1 << d) + t] = e;
            }
            T::wave_sync();
        }
        for (int d = depth; d >= 0; --d) {
            DP mine = (d & 1) ? leafbuf + 128 : acc8;
            DP below = (d & 1) ? acc8 : leafbuf + 128;
            for (int t = T::lane(); t < (1 << d); t += T::WSIZE) {
                const int e = tab[(1 << d) + t];
                const int off = e & 0xffff, l = e >> 16;
                if (l > 0) mine[t] = l <= 128 ? leafbuf[off >> 6] : below[2 * t] + below[2 * t + 1];
            }
            T::wave_sync();
        }
    }
    T::sync();
    return acc8[0];
}
template <class T>
PW_HD inline double np_walk_phase(int len, int* tab, double* acc8, double* leafbuf) {
    if (PW_IS_LDS(acc8) && PW_IS_LDS(leafbuf) && PW_IS_LDS(tab)) return np_walk_phase_t<T>(len, PW_AS_LDS(tab), PW_AS_LDS(acc8), PW_AS_LDS(leafbuf));
    return np_walk_phase_t<T>(len, tab, acc8, leafbuf);
}
template <class T>
PW_HD inline double np_sum_team(const double* a, int n, int* tab, double* acc8, double* leafbuf,
                                double* slot) {
    double total = 0.0;
    bool first = true;
    for (int s = 0; s < n; s += 8192) {
        const int len = n - s < 8192 ? n - s : 8192;
        np_leaf_phase<T>(a + s, len, 0, len, tab, acc8, leafbuf);
        double part = np_walk_phase<T>(len, tab, acc8, leafbuf);
        if (T::tid() == 0) total = first ? part : total + part;
        first = false;
        T::sync();
    }
    if (T::tid() == 0) *slot = total;
    T::sync();
    double r = *slot;
    T::sync();
    return r;
}

// ---- sampling sphere -----------------------------------------------------------------
struct Sphere {
    int P;
    double R, start, stop, step;
    PW_HD void init(double radius, int count) {
        P = count;
        R = radius;
        start = 1.0 - 1.0 / (double)count;
        stop = 1.0 / (double)count - 1.0;
        step = (stop - start) / (double)(count - 1);
    }
    PW_HD double zunit(int k) const { return (k == P - 1) ? stop : (double)k * step + start; }
    PW_HD void point(int k, double* px, double* py, double* pz) const {
        double theta = GOLDEN_ANGLE * (double)k;
        double z = zunit(k);
        double ring = pw_sqrt(1.0 - z * z);
        double s, c;
        pw_sincos(theta, &s, &c);
        *px = ring * c * R;
        *py = ring * s * R;
        *pz = z * R;
    }
};
// ---- neighbour tables of the sampling sphere ----------------------------------------------------
// The DBSCAN radius is the mean of the ten nearest-neighbour distances of ALL P sampling vectors
// (utilities.py:1427-1434).  The vectors are a golden spiral scaled by the sphere radius R, so WHICH
// points are a point's nearest neighbours is a property of (P, point) alone; only the distances depend on
// R.  For every P of interest the sixteen nearest points of each point ON THE UNIT SPHERE are tabulated
// once per context (itself included, ascending), together with the squared unit distance of the
// seventeenth -- a lower bound for every point not in the list.  At run time the sixteen exact distances
// (the reference's arithmetic, on the scaled vectors) are formed; if they come out ascending and the tenth
// is provably below everything outside the list, the first ten ARE the reference's ten -- otherwise (exact
// ties, never seen) the point goes through the windowed search.  16 distances per point instead of ~200.
constexpr int PW_NB_K = 16;
constexpr int PW_NB_PMIN = 32;
constexpr int PW_NB_PMAX = PW_P_MAX;
constexpr unsigned PW_NB_NONE = 0xffffffffu;
// first point of the table of P when the tables of PW_NB_PMIN .. P - 1 precede it
PW_HD inline unsigned nb_dense_offset(int P) {
    return (unsigned)(((long)(P - 1) * P - (long)(PW_NB_PMIN - 1) * PW_NB_PMIN) / 2);
}
// The sampling vectors of a sphere of radius R with P points, handed to store(k, x, y, z) by the team.  Sphere::point is
// (ring * cos) * R, (ring * sin) * R, z * R: with the context's table of unit vectors (Sphere::point at R = 1, written
// by the kernel that builds the neighbour tables) a vector is three loads and three products -- the same bits, since
// x * 1.0 is x -- instead of a sine and a cosine of k golden angles in numpy's arithmetic (a third of the stage before
// the rays, for every unit of every launch that samples a sphere).  No table (host path, PW_NB_TABLES=0): computed.
template <class T, class ST>
PW_HD inline __attribute__((always_inline)) void team_sphere_points(const TeamWorkspace* ws, const Sphere& sp, ST store) {
    const int P = sp.P;
    const double* u = nullptr;
    if (ws->nb_unit && ws->nb_off && P >= PW_NB_PMIN && P <= PW_NB_PMAX && ws->nb_off[P] != PW_NB_NONE)
        u = ws->nb_unit + 3 * (size_t)ws->nb_off[P];
    for (int k = T::tid(); k < P; k += T::SIZE) {
        double x, y, z;
        if (u) { x = u[3 * k] * sp.R; y = u[3 * k + 1] * sp.R; z = u[3 * k + 2] * sp.R; }
        else sp.point(k, &x, &y, &z);
        store(k, x, y, z);
    }
}
// one row of a table: the PW_NB_K nearest of point k among the P unit vectors u*, and the bound
PW_HD inline void nb_build_point(int P, int k, const double* ux, const double* uy, const double* uz,
                                 unsigned short* idx, double* bound) {
    double t[PW_NB_K + 1];
    int id[PW_NB_K + 1];
#pragma unroll
    for (int q = 0; q <= PW_NB_K; ++q) { t[q] = PW_INF; id[q] = 0; }
    const double x = ux[k], y = uy[k], z = uz[k];
    for (int j = 0; j < P; ++j) {
        double ax = x - ux[j], ay = y - uy[j], az = z - uz[j];
        double d = ax * ax + ay * ay + az * az;
        if (d < t[PW_NB_K]) {
            int jj = j;
#pragma unroll
            for (int q = 0; q <= PW_NB_K; ++q) {
                const bool lower = d < t[q];
                const double td = t[q];
                const int ti = id[q];
                t[q] = lower ? d : td; id[q] = lower ? jj : ti;
                d = lower ? td : d; jj = lower ? ti : jj;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PW_NB_K; ++q) idx[q] = (unsigned short)id[q];
    *bound = t[PW_NB_K];
}

PW_HD inline pw_params default_params() {
    pw_params p;
    p.adjust_windows = 1.0; p.adjust_average = 1.0; p.increment = 1.0; p.pore_opt = 1; p.opt_flags = 0;
    for (int c = 0; c < 3; ++c) { p.opt_x0[c] = 0.0; p.opt_lo[c] = -PW_INF; p.opt_hi[c] = PW_INF; }
    p.increment2 = 0.1; p.z_lo = -PW_INF; p.z_hi = PW_INF; p.lb_z = 1; p.z_second_mini = 0;
    return p;
}
// int(np.log10(4*pi*r**2) * 250 * adjust)  (utilities.py:1410, 1615)
PW_HD inline int sampling_count(double radius, double adjust) {
    double area = FOUR_PI * pw_square_np(radius);    // radius ** 2 on a float scalar: libm pow
    return (int)((pw_log10(area) * 250.0) * adjust);
}

// Ray from the centroid along (dx,dy,dz) against every atom (utilities.py:1138-1158 /
// 1561-1578).  Returns whether any atom is "in the way" and the largest |p_out|.
PW_HD inline __attribute__((always_inline)) bool ray_scan_impl(const Frame& F, int n, const double* cen, double dx, double dy,
                           double dz, double* farthest) {
    double nrm = norm3(dx, dy, dz);
    double ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    bool any = false;
    double far = -1.0;
    // Two steps per block of 64 atoms: a cheap conservative screen (no square roots)
    // marks the atoms whose sphere the line can touch; only those go through the
    // reference's exact arithmetic.  Screen: the reference needs
    // vdw^2 - fl(sqrt(q))^2 > 0 with q = |rel|^2 - along^2 (NaN for q < 0 => no hit);
    // q > vdw^2 (1 + 1e-14) makes that impossible whatever the rounding.
    for (int blk = 0; blk < n; blk += 64) {
        unsigned long long mask = 0;
        int jend = n - blk < 64 ? n - blk : 64;
        int j = 0;
        for (; j + 8 <= jend; j += 8) {
            double ax[8], ay[8], az[8], ar[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                int i = blk + j + t;
                ax[t] = F.x[i]; ay[t] = F.y[i]; az[t] = F.z[i]; ar[t] = F.vdw[i];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                double rx = ax[t] - cen[0], ry = ay[t] - cen[1], rz = az[t] - cen[2];
                double along = pw_fma(rz, uz, pw_fma(rx, ux, ry * uy));
                double q = sq3(rx, ry, rz) - along * along;
                double r2 = ar[t] * ar[t];
                if (q >= 0.0 && q <= r2 * (1.0 + 1e-14)) mask |= 1ull << (j + t);
            }
        }
        for (; j < jend; ++j) {
            int i = blk + j;
            double rx = F.x[i] - cen[0], ry = F.y[i] - cen[1], rz = F.z[i] - cen[2];
            double along = pw_fma(rz, uz, pw_fma(rx, ux, ry * uy));
            double q = sq3(rx, ry, rz) - along * along;
            double r2 = F.vdw[i] * F.vdw[i];
            if (q >= 0.0 && q <= r2 * (1.0 + 1e-14)) mask |= 1ull << j;
        }
        while (mask) {
            int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            int i = blk + j;
            double rx = F.x[i] - cen[0], ry = F.y[i] - cen[1], rz = F.z[i] - cen[2];
            double along = pw_fma(rz, uz, pw_fma(rx, ux, ry * uy));
            double sq = sq3(rx, ry, rz);
            double perp = pw_sqrt(sq - along * along);
            double radicand = F.vdw[i] * F.vdw[i] - perp * perp;
            if (radicand > 0.0) {
                double half = pw_sqrt(radicand);
                double tin = along - half, tout = along + half;
                double ix = cen[0] + tin * ux, iy = cen[1] + tin * uy, iz = cen[2] + tin * uz;
                double ox = cen[0] + tout * ux, oy = cen[1] + tout * uy, oz = cen[2] + tout * uz;
                double nin = norm3(ix, iy, iz), nout = norm3(ox, oy, oz);
                if (nin < nout) {
                    any = true;
                    if (nout > far) far = nout;
                }
            }
        }
    }
    *farthest = far;
    return any;
}

PW_NOINLINE PW_HD inline bool ray_scan(const Frame& F, int n, const double* cen, double dx, double dy, double dz,
                                       double* farthest) {
    return ray_scan_impl(F, n, cen, dx, dy, dz, farthest);
}

// NR (= 4) rays of one thread at once: the screen reads every atom once for all of them (register
// blocking); the exact part per flagged (ray, atom) is the one of ray_scan_impl.  Results are
// identical to NR calls of ray_scan.  FAR = false: only `hit` is wanted (the sampling stage of
// find_windows), so a ray that has hit something drops its remaining candidates.
template <int NR, bool FAR>
PW_HD inline __attribute__((always_inline)) void ray_scan_multi_impl(const Frame& F, int n, const double* cen,
                                                                     const double* dx, const double* dy,
                                                                     const double* dz, bool* hit, double* farthest) {
    static_assert(NR == 4, "the merged exact loop below is written for four rays");
    double ux[NR], uy[NR], uz[NR], far[NR];
    bool any[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        double nrm = norm3(dx[r], dy[r], dz[r]);
        ux[r] = dx[r] / nrm; uy[r] = dy[r] / nrm; uz[r] = dz[r] / nrm;
        far[r] = -1.0;
        any[r] = false;
    }
    const double c0 = cen[0], c1 = cen[1], c2 = cen[2];
    // The screen only has to be conservative, so it is as short as it can be: the squared distance
    // from the line with one fused operation, one comparison.  It differs from the value the exact
    // code below forms (|rel|^2 - along^2 rounded twice) by less than 2e-14 for these magnitudes; the
    // limit carries 1e-12 (relative to the radius and to |rel|^2), and an atom whose value comes out
    // negative is flagged too -- the exact code finds the NaN there, as the reference does.
    // The flagged (ray, atom) pairs of 64 atoms are then worked off in ONE loop over all four rays: a
    // wave runs a loop as long as its busiest lane, and the busiest lane of a loop per (ray, 32 atoms)
    // has three pairs where the average lane has 0.6 -- merged, the ratio is two.
    for (int blk = 0; blk < n; blk += 64) {
        unsigned long long mask[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) mask[r] = 0ull;
        for (int half = 0; half < 64 && blk + half < n; half += 32) {
            unsigned mh[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) mh[r] = 0u;
            const int base = blk + half;
            const int jend = n - base < 32 ? n - base : 32;
PW_PRAGMA(unroll PW_UNROLL_RAY)
            for (int j = 0; j < jend; ++j) {
                int i = base + j;
                double rx = F.x[i] - c0, ry = F.y[i] - c1, rz = F.z[i] - c2;
                double rr = sq3(rx, ry, rz);
                double lim = pw_fma(rr, 1e-12, (F.vdw[i] * F.vdw[i]) * (1.0 + 1e-12));
                const unsigned bit = 1u << j;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    double along = pw_fma(rz, uz[r], pw_fma(rx, ux[r], ry * uy[r]));
                    double q = pw_fma(-along, along, rr);
                    mh[r] |= q <= lim ? bit : 0u;
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) mask[r] |= (unsigned long long)mh[r] << half;
        }
        if (!FAR) {
#pragma unroll
            for (int r = 0; r < NR; ++r) mask[r] = any[r] ? 0ull : mask[r];
        }
        while ((mask[0] | mask[1] | mask[2] | mask[3]) != 0ull) {
            // the next pair of this thread: the lowest flagged atom of its first ray that has one
            const int r = mask[0] ? 0 : (mask[1] ? 1 : (mask[2] ? 2 : 3));
            const unsigned long long m = r == 0 ? mask[0] : (r == 1 ? mask[1] : (r == 2 ? mask[2] : mask[3]));
            const unsigned long long rest = m & (m - 1);
            const int i = blk + __builtin_ctzll(m);
            const double vx = r == 0 ? ux[0] : (r == 1 ? ux[1] : (r == 2 ? ux[2] : ux[3]));
            const double vy = r == 0 ? uy[0] : (r == 1 ? uy[1] : (r == 2 ? uy[2] : uy[3]));
            const double vz = r == 0 ? uz[0] : (r == 1 ? uz[1] : (r == 2 ? uz[2] : uz[3]));
            bool got = false;
            double nout = -1.0;
            double rx = F.x[i] - c0, ry = F.y[i] - c1, rz = F.z[i] - c2;
            double along = pw_fma(rz, vz, pw_fma(rx, vx, ry * vy));
            double sq = sq3(rx, ry, rz);
            double perp = pw_sqrt(sq - along * along);
            double radicand = F.vdw[i] * F.vdw[i] - perp * perp;
            if (radicand > 0.0) {
                double half = pw_sqrt(radicand);
                double tin = along - half, tout = along + half;
                double ix = c0 + tin * vx, iy = c1 + tin * vy, iz = c2 + tin * vz;
                double ox = c0 + tout * vx, oy = c1 + tout * vy, oz = c2 + tout * vz;
                double nin = norm3(ix, iy, iz);
                nout = norm3(ox, oy, oz);
                got = nin < nout;
            }
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const bool mine = r == q;
                if (mine) mask[q] = (!FAR && got) ? 0ull : rest;
                if (mine && got) {
                    any[q] = true;
                    if (FAR && nout > far[q]) far[q] = nout;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) { hit[r] = any[r]; farthest[r] = far[r]; }
}
template <int NR, bool FAR = true>
PW_NOINLINE PW_HD inline void ray_scan_multi(const Frame& F, int n, const double* cen, const double* dx,
                                             const double* dy, const double* dz, bool* hit, double* farthest) {
    ray_scan_multi_impl<NR, FAR>(F, n, cen, dx, dy, dz, hit, farthest);
}


// ---- ray tests of a whole sampling sphere, atom by atom --------------------------------------------
// vector_preanalysis / vector_analysis_reversed (utilities.py:1132-1161, 1556-1583) ask, for each of P
// rays from the centroid, which atoms the line meets.  An atom of radius r at distance |rel| can only be
// met by rays inside a cone of half-angle asin(r / |rel|) about its direction -- some 1 % of the sphere --
// and the rays of a golden spiral with a given polar angle are a contiguous index range.  So instead of
// testing every (ray, atom) pair (P x N screens), one wave per atom walks the atom's index range, keeps
// the rays inside the cone (one dot product each, conservative) as (ray, atom) pairs, and the exact
// arithmetic of the reference then runs on the pairs only, one pair per lane.  A ray is stopped if any of
// its pairs says so and its farthest exit is the maximum over its pairs: both independent of the order in
// which the pairs are visited, so the results are those of the dense scan, bit for bit.
// Returns false (nothing written) when the pair list is too small or an index does not fit 16 bits.
PW_HD inline void team_store_max_pos(double* p, double v) {     // *p = max(*p, v) for v > 0, *p >= -1
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax((long long*)p, __double_as_longlong(v));           // (positive doubles order like their bits)
#else
    if (v > *p) *p = v;
#endif
}
// How team_ray_tests reads ray vector k: `first(k)` is where its x component sits, `at(off, ...)` reads the three
// components from there, and STEP64 is what 64 rays further on adds to `off` -- the band walk computes the offset of a
// lane's first ray once per atom instead of the layout's index arithmetic once per ray.  P: the pointer type (a
// team-memory pointer where the vectors are known to be there: ds_read instead of flat_load).
template <class P>
struct SpiralRays {             // the window search's layout: component c of ray k at [(c * 4 + (k & 3)) * Q4 + (k >> 2)]
    P pts;
    int Q4;
    static constexpr int STEP64 = 16;
    PW_HD int first(int k) const { return (k & 3) * Q4 + (k >> 2); }
    PW_HD void at(int off, double* x, double* y, double* z) const { *x = pts[off]; *y = pts[off + 4 * Q4]; *z = pts[off + 8 * Q4]; }
    PW_HD void operator()(int k, double* x, double* y, double* z) const { at(first(k), x, y, z); }
};
template <class P>
struct PlanarRays {             // the average diameter's layout: x | y | z, P each
    P a;
    int n;
    static constexpr int STEP64 = 64;
    PW_HD int first(int k) const { return k; }
    PW_HD void at(int off, double* x, double* y, double* z) const { *x = a[off]; *y = a[n + off]; *z = a[2 * n + off]; }
    PW_HD void operator()(int k, double* x, double* y, double* z) const { at(k, x, y, z); }
};
// per-atom cone data computed by one lane each (band of ray indices, threshold on dot(ray vector, rel))
struct ConeBand { double thr; int klo, khi; };     // khi < 0: two-sided test (|dot|), khi = -khi - 1
template <class T, bool FAR, class GETP>
PW_HD inline __attribute__((always_inline)) bool team_ray_tests(const Frame& F, int n, const double* cen, const Sphere& sp,
                                                               GETP getp, ConeBand* bands, unsigned* pairs, int cap,
                                                               PW_LDS int* counts, unsigned char* flag,
                                                               unsigned char hit_value, double* far) {
    const int P = sp.P;
    const int seg = cap / T::NWAVES;                 // every wave appends to a segment of its own: no atomics
    if (P >= 65536 || n >= 65536 || seg < 64) return false;
    const double c0 = cen[0], c1 = cen[1], c2 = cen[2];
    const double cn = norm3(c0, c1, c2);
    // ---- the cones, one atom per lane ----
    for (int i = T::tid(); i < n; i += T::SIZE) {
        const double rx = F.x[i] - c0, ry = F.y[i] - c1, rz = F.z[i] - c2, vr = F.vdw[i];
        const double rr = sq3(rx, ry, rz);
        // the line meets the sphere only if |rel|^2 - along^2 <= r^2 (with the margin of the dense screen)
        const double a2 = rr - pw_fma(rr, 1e-12, (vr * vr) * (1.0 + 1e-12));
        ConeBand b;
        b.klo = 0; b.khi = -P;                     // every ray, two-sided
        b.thr = -PW_INF;
        if (a2 > 0.0) {
            const double amin = pw_sqrt(a2) * (1.0 - 1e-9);      // smallest |along| of a meeting line
            b.thr = amin * sp.R * (1.0 - 1e-9);
            // an atom behind the centroid (along < 0) only counts if along > -(cen . u) >= -|cen|
            // (utilities.py:1152-1155 compares distances from the ORIGIN): out of the question when the
            // cone's smallest |along| exceeds |cen|
            if (amin > cn + 1e-6) {
                const double len = pw_sqrt(rr);
                const double zd = rz / len, c = amin / len;
                const double s_ = pw_sqrt(pw_max(1.0 - c * c, 0.0)) * (1.0 + 1e-9) + 1e-9;
                const double rho = pw_sqrt(pw_max(1.0 - zd * zd, 0.0));
                const double zhi = zd >= c ? 1.0 : pw_min(1.0, zd * c + rho * s_ + 1e-9);
                const double zlo = zd <= -c ? -1.0 : pw_max(-1.0, zd * c - rho * s_ - 1e-9);
                // z_k = start + k * step, step < 0
                const double kh = (zlo - sp.start) / sp.step, kl = (zhi - sp.start) / sp.step;
                b.klo = (int)pw_max(kl - 2.0, 0.0);
                b.khi = (int)pw_min(kh + 3.0, (double)(P - 1));
            }
        }
        bands[i] = b;
    }
    T::sync();
    // ---- the rays inside each cone: one wave per atom walks the atom's index band ----
    int mine = 0;                                   // pairs appended by this wave so far (wave-uniform)
    unsigned* my = pairs + (size_t)T::wave() * seg;
    for (int i = T::wave(); i < n; i += T::NWAVES) {
        const ConeBand b = bands[i];
        const bool two_sided = b.khi < 0;
        const int khi = two_sided ? -b.khi - 1 : b.khi;
        PW_DCHECK(P >= 10 && P <= 4096, 102);
        PW_DCHECK(b.klo >= 0 && khi < P, 103);
        const double rx = F.x[i] - c0, ry = F.y[i] - c1, rz = F.z[i] - c2;
        int off = getp.first(b.klo + T::lane());
        for (int kb = b.klo; kb <= khi; kb += T::WSIZE, off += GETP::STEP64) {
            const int k = kb + T::lane();
            bool f = false;
            if (k <= khi) {
                double px, py, pz;
                if (T::WSIZE == 64) getp.at(off, &px, &py, &pz);      // (the offsets step by 64 rays)
                else getp(k, &px, &py, &pz);
                const double dot = pw_fma(pz, rz, pw_fma(px, rx, py * ry));
                f = two_sided ? pw_abs(dot) >= b.thr : dot >= b.thr;
            }
            const unsigned long long bal = T::ballot(f);
            const int pos = mine + __builtin_popcountll(bal & ((1ull << T::lane()) - 1ull));
            if (f && pos < seg) my[pos] = ((unsigned)k << 16) | (unsigned)i;
            mine += __builtin_popcountll(bal);
        }
    }
    if (T::lane() == 0) counts[T::wave()] = mine;
    T::sync();
    int cum[T::NWAVES + 1];
    cum[0] = 0;
    bool fits = true;
#pragma unroll
    for (int w = 0; w < T::NWAVES; ++w) { const int cw = counts[w]; fits = fits && cw <= seg; cum[w + 1] = cum[w] + cw; }
    T::sync();
    if (!fits) return false;
    // ---- the reference's arithmetic on the pairs, one pair per lane ----
    for (int g = T::tid(); g < cum[T::NWAVES]; g += T::SIZE) {
        int w = 0, first = 0;                   // (selects, not an indexed read of cum[]: that would live in scratch)
#pragma unroll
        for (int q = 1; q < T::NWAVES; ++q)
            if (g >= cum[q]) { w = q; first = cum[q]; }
        const unsigned pr = pairs[(size_t)w * seg + (g - first)];
        const int k = (int)(pr >> 16), i = (int)(pr & 0xffffu);
        double dx, dy, dz;
        getp(k, &dx, &dy, &dz);
        const double nrm = norm3(dx, dy, dz);
        const double ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
        const double rx = F.x[i] - c0, ry = F.y[i] - c1, rz = F.z[i] - c2;
        const double along = pw_fma(rz, uz, pw_fma(rx, ux, ry * uy));
        const double sq = sq3(rx, ry, rz);
        const double perp = pw_sqrt(sq - along * along);
        const double radicand = F.vdw[i] * F.vdw[i] - perp * perp;
        if (radicand > 0.0) {
            const double half = pw_sqrt(radicand);
            const double tin = along - half, tout = along + half;
            const double ix = c0 + tin * ux, iy = c1 + tin * uy, iz = c2 + tin * uz;
            const double ox = c0 + tout * ux, oy = c1 + tout * uy, oz = c2 + tout * uz;
            const double nin = norm3(ix, iy, iz), nout = norm3(ox, oy, oz);
            if (nin < nout) {
                flag[k] = hit_value;
                if (FAR) team_store_max_pos(&far[k], nout);
            }
        }
    }
    T::sync();
    return true;
}

// numpy floor division a // b for positive doubles (npy_divmod)
PW_HD inline double np_floordiv(double a, double b) {
    double mod = __builtin_fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0.0) != (mod < 0.0)) div -= 1.0;
    }
    double fl;
    if (div != 0.0) {
        fl = __builtin_floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else {
        fl = 0.0;
    }
    return fl;
}

// vector_analysis (utilities.py:1100-1129) by ONE thread: walk 0 -> v.
// returns false if some point is inside a vdW sphere.
// m0: the gap at the origin.  Every path starts there (its first point is 0 * step), so the caller evaluates it once
// for all of them and the walk begins at the second point; the points that are left go six to a pass over the atoms,
// the last pass five when no more are left (eleven points -- the usual count -- are 6 + 5, not two passes of six).
PW_NOINLINE PW_HD inline bool path_scan_thread(const Frame& F, int n, double vx, double vy, double vz,
                                   double inc, double m0, double* out_2gap, int* out_pos, double* out_chunk,
                                   int* n_eval, const lint* cand = nullptr, const lint* coff = nullptr) {
    double nrm = norm3(vx, vy, vz);
    int chunks = (int)np_floordiv(nrm, inc);
    double cx = vx / (double)chunks, cy = vy / (double)chunks, cz = vz / (double)chunks;
    double best = PW_INF;
    int pos = 0;
    bool ok = true;
    int k0 = 0;
    if (chunks >= 1) {
        // (the steps are finite: point 0 is the origin exactly, whatever the signs of its zeros)
        if (!(m0 > 0.0)) ok = false;
        best = m0;
        k0 = 1;
    }
    // NP path points per pass over the atoms (the reference stops at the first point inside a
    // sphere; evaluating the rest changes nothing: the vector is rejected either way)
    auto tile = [&](auto np_) __attribute__((always_inline)) {
        constexpr int NP = decltype(np_)::value;
        double qx[NP], qy[NP], qz[NP], m[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int k = k0 + p <= chunks ? k0 + p : chunks;
            qx[p] = cx * (double)k; qy[p] = cy * (double)k; qz[p] = cz * (double)k;
        }
        points_gap_values<NP>(F, n, qx, qy, qz, m, cand, coff);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (k0 + p <= chunks && ok) {
                if (!(m[p] > 0.0)) ok = false;
                else if (m[p] < best) { best = m[p]; pos = k0 + p; }
            }
        }
        k0 += NP;
    };
    while (k0 <= chunks && ok) {
        if (PW_TILE_PATH > 5 && chunks - k0 + 1 <= 5) tile(std::integral_constant<int, 5>());
        else tile(std::integral_constant<int, PW_TILE_PATH>());
    }
    if (n_eval) *n_eval += chunks + 1;
    if (!ok) return false;
    *out_2gap = best * 2.0;
    *out_pos = pos;
    out_chunk[0] = cx; out_chunk[1] = cy; out_chunk[2] = cz;
    return true;
}

// The atoms that can matter to the paths of ONE wave (a path each in the lanes that `have` one; vx, vy, vz its end point).
// A path's result is the smallest gap over its points and where it is first reached, and it starts at m0, the gap at
// the origin (path_scan_thread): an atom whose gap is PROVABLY above m0 at every point but the origin of every path of
// the wave changes nothing -- not the minimum, not its place, not the test for a point inside a sphere (that needs a gap
// <= 0 < m0).  The wave's directions lie in a band of latitudes z1 <= u_z <= z2; the angle between an atom's direction and
// ANY direction of the band is at least its distance in latitude to the band, theta1.  With rho the atom's distance from
// the origin, c1 = cos(theta1) (0 beyond a right angle) and t1 = inc <= the distance of a path's second point from the origin, the
// squared distance from the atom to a point at t >= t1 along such a direction is at least rho^2 + t^2 - 2 t rho c1, i.e.
// at least rho^2 (1 - c1^2) where rho c1 >= t1 and rho^2 + t1^2 - 2 t1 rho c1 otherwise.  An atom goes on the list unless
// that bound exceeds (r + m0)^2, margins of 1e-9 on every side (the quantities are good to 1e-15).  The list is in stored
// order (so the groups by radius stay contiguous): cand[0 .. coff[k]), group g at [coff[g], coff[g + 1]).  Lanes test
// atoms, 64 at a time: a hundred instructions for a wave's list.  false: no list (ungrouped radii, a path without a
// second point, m0 <= 0 -- every path fails at the origin then): the caller scans all atoms.
template <class T>
PW_HD inline bool wave_path_candidates(const Frame& F, int n, bool have, double vx, double vy, double vz, double inc,
                                       double m0, lint* cand, lint* coff) {
    const auto& C = *F.cls;
    if (C.k == 0 || !(m0 > 0.0)) return false;
    double uz_lo = PW_INF, uz_hi = -PW_INF, t1 = PW_INF;
    bool odd = false;
    if (have) {
        // (a path of at least one step: its length / floor(length / inc) is at least inc -- no floor division here)
        const double nrm = norm3(vx, vy, vz);
        odd = !(nrm >= inc * (1.0 + 1e-9)) || !(nrm < PW_INF) || !(inc > 0.0);
        if (!odd) uz_lo = uz_hi = vz / nrm;
    }
    if (T::wave_any(odd)) return false;
    double z1 = T::wave_min(uz_lo) - 1e-9, z2 = -T::wave_min(-uz_hi) + 1e-9;
    t1 = inc * (1.0 - 1e-9);
    if (!(z1 <= z2)) return false;               // (no lane has a path)
    z1 = pw_max(z1, -1.0); z2 = pw_min(z2, 1.0);
    const double s1 = pw_sqrt(pw_max(1.0 - z1 * z1, 0.0)), s2 = pw_sqrt(pw_max(1.0 - z2 * z2, 0.0));
    const double m0p = m0 * (1.0 + 1e-9) + 1e-9;
    // lane g counts the listed atoms below group g's first position: coff[g]
    const int mycut = T::lane() <= C.k ? C.off[T::lane() <= PW_KCLS ? T::lane() : PW_KCLS] : n;
    int mycount = 0, cnt = 0;
    for (int base = 0; base < n; base += T::WSIZE) {
        const int i = base + T::lane();
        bool keep = false;
        if (i < n) {
            const double xx = F.xx[i], z = F.z[i], r = F.vdw[i];
            const double rho = pw_sqrt(xx);
            const double zd = rho > 0.0 ? z / rho : 0.0;
            const double sd = pw_sqrt(pw_max(1.0 - zd * zd, 0.0));
            double c1 = 1.0;
            if (zd > z2) c1 = sd * s2 + zd * z2;
            else if (zd < z1) c1 = sd * s1 + zd * z1;
            c1 = pw_min(1.0, pw_max(0.0, c1 + 1e-9));
            const double ts = rho * c1;
            const double low = ts >= t1 ? xx * (1.0 - c1 * c1) : (xx + t1 * t1) - 2.0 * t1 * ts;
            const double lim = (r + m0p) * (r + m0p);
            keep = !(low * (1.0 - 1e-9) - 1e-9 > lim * (1.0 + 1e-9));
        }
        const unsigned long long bal = T::ballot(keep);
        const int pos = cnt + __builtin_popcountll(bal & ((1ull << T::lane()) - 1ull));
        if (keep) cand[pos] = i;
        if (T::WSIZE > 1) {
            int w = mycut - base;
            w = w < 0 ? 0 : (w > 64 ? 64 : w);
            mycount += __builtin_popcountll(bal & (w >= 64 ? ~0ull : ((1ull << w) - 1ull)));
        }
        cnt += __builtin_popcountll(bal);
    }
    if (T::WSIZE > 1) {
        if (T::lane() <= C.k) coff[T::lane()] = mycount;
    } else {
        int c = 0;
        for (int g = 0; g <= C.k; ++g) {
            while (c < cnt && cand[c] < C.off[g]) ++c;
            coff[g] = c;
        }
    }
    T::wave_sync();
    return true;
}

// ---- stage: load --------------------------------------------------------------------------
// The grouping of the atoms by radius is a property of the RADII: in a batch of one molecule type (a trajectory: one
// template of radii for every unit) it is the same for every unit, and working it out per unit -- every atom against
// every other, twice -- was 35 us of a chain's wave and 10 us of a window team (in-kernel timer, round 6).  For such a
// batch the host works it out once (template_groups_build, the statements of load_unit below) and the launches read it:
// a blob of ClassInfo | inv[npad] | perm[npad] | radii in stored order [npad] | the molecular weight (numpy's sum of the
// masses: the same for every unit as well), npad = n rounded up to even.
PW_HD inline size_t template_groups_bytes(int n) {
    const size_t npad = (size_t)((n + 1) & ~1);
    return ((sizeof(ClassInfo) + 7) & ~(size_t)7) + npad * 4 * 2 + npad * 8 + 8;
}
// numpy's sum of the masses (np.sum over a contiguous vector: pairwise, chunks of 8192)
PW_HD inline double molecular_weight_sum(const double* mass, int n) {
    double tot = 0.0;
    for (int s0 = 0; s0 < n; s0 += 8192) {
        // (np_sum_lean: the recursion walked with scalars only -- np_sum_small's leaf tables were 500 bytes of
        // scratch per lane in every kernel that contains this stage)
        double part = np_sum_lean(mass + s0, n - s0 < 8192 ? n - s0 : 8192);
        tot = s0 == 0 ? part : tot + part;
    }
    return tot;
}
inline void template_groups_build(const double* vdw, const double* mass, int n, unsigned char* blob) {
    const size_t npad = (size_t)((n + 1) & ~1);
    ClassInfo* cls = (ClassInfo*)blob;
    int* inv = (int*)(blob + ((sizeof(ClassInfo) + 7) & ~(size_t)7));
    int* perm = inv + npad;
    double* vs = (double*)(perm + npad);
    memset(blob, 0, template_groups_bytes(n));
    // key_i = first atom with the same radius; groups in order of first appearance, stable inside a group
    int ngrp = 0;
    for (int i = 0; i < n; ++i) {
        int key = i;
        for (int j = 0; j < i; ++j)
            if (vdw[j] == vdw[i]) { key = j; break; }
        perm[i] = key;                                   // (keys, until the scatter below)
        if (key == i) ++ngrp;
    }
    if (ngrp > PW_KCLS) {
        cls->k = 0; cls->off[0] = 0;
        for (int i = 0; i < n; ++i) inv[i] = i;
    } else {
        cls->k = ngrp;
        for (int i = 0; i < n; ++i) {
            const int key = perm[i];
            int pos = 0, cid = 0;
            for (int j = 0; j < n; ++j) {
                const int kj = perm[j];
                pos += (kj < key) || (kj == key && j < i);
                cid += (kj == j) && (j < key);
            }
            inv[i] = pos;
            if (key == i) { cls->vdw[cid] = vdw[i]; cls->off[cid] = pos; }
        }
        cls->off[ngrp] = n;
    }
    for (int i = 0; i < n; ++i) { vs[inv[i]] = vdw[i]; }
    for (int i = 0; i < n; ++i) perm[inv[i]] = i;       // (every key has been read)
    vs[npad] = molecular_weight_sum(mass, n);
}

template <class T>
PW_HD inline void load_unit(UnitShared& sh, int n, const double* xyz, const double* vdw,
                            const double* mass, const unsigned char* tmpl = nullptr) {
    auto& v = *sh.v;
    if (tmpl) {
        // the grouping comes with the batch (see above): one pass, one barrier
        const size_t npad = (size_t)((n + 1) & ~1);
        const ClassInfo* tc = (const ClassInfo*)tmpl;
        const int* t_inv = (const int*)(tmpl + ((sizeof(ClassInfo) + 7) & ~(size_t)7));
        const int* t_perm = t_inv + npad;
        const double* t_vs = (const double*)(t_perm + npad);
        for (int i = T::tid(); i < n; i += T::SIZE) {
            const int pos = t_inv[i];
            const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            sh.A.x[pos] = x; sh.A.y[pos] = y; sh.A.z[pos] = z;
            sh.A.xx[pos] = sq3(x, y, z);
            sh.inv[i] = pos;
            sh.perm[i] = t_perm[i];
            sh.vdw[i] = t_vs[i];
            sh.mass[i] = mass[i];
        }
        if (T::tid() == 0) {
            v.n_eval = 0; v.status = 0;
            v.mw = t_vs[npad]; v.mw_given = 1;
            v.cls.k = tc->k;
            for (int g = 0; g <= PW_KCLS; ++g) v.cls.off[g] = tc->off[g];
            for (int g = 0; g < PW_KCLS; ++g) v.cls.vdw[g] = tc->vdw[g];
        }
        T::sync();
        return;
    }
    // Group atoms by radius, stable (ascending atom index inside a group), groups in order of
    // first appearance -- computed by every thread for its own atoms:
    //   key_i  = index of the first atom with the same radius
    //   pos_i  = #{j : key_j < key_i} + #{j < i : key_j == key_i}
    for (int i = T::tid(); i < n; i += T::SIZE) {
        double r = vdw[i];
        sh.vdw[i] = r;                 // caller's order for the moment
        sh.mass[i] = mass[i];
    }
    if (T::tid() == 0) { v.n_eval = 0; v.status = 0; v.cls.k = 0; v.mw_given = 0; }
    T::sync();
    for (int i = T::tid(); i < n; i += T::SIZE) {
        double r = sh.vdw[i];
        int key = i;
        for (int j = 0; j < i; ++j)
            if (sh.vdw[j] == r) { key = j; break; }
        sh.perm[i] = key;              // perm[] holds the keys until the scatter below
    }
    T::sync();
    for (int i = T::tid(); i < n; i += T::SIZE) {
        int key = sh.perm[i];
        int pos = 0, cid = 0;
        for (int j = 0; j < n; ++j) {
            int kj = sh.perm[j];
            pos += (kj < key) || (kj == key && j < i);
            cid += (kj == j) && (j < key);      // group leaders before mine
        }
        sh.inv[i] = pos;
        if (key == i) {                // group leader: publishes its group
            if (cid < PW_KCLS) { v.cls.vdw[cid] = sh.vdw[i]; v.cls.off[cid] = pos; }
            team_atomic_max(&v.cls.k, cid + 1);
        }
    }
    T::sync();
    const int ngrp = v.cls.k;
    T::sync();
    if (ngrp > PW_KCLS) {
        // too many distinct radii: no grouping (bulk evaluations fall back to per-atom sqrt)
        for (int i = T::tid(); i < n; i += T::SIZE) sh.inv[i] = i;
        if (T::tid() == 0) { v.cls.k = 0; v.cls.off[0] = 0; }
    } else if (T::tid() == 0) {
        v.cls.off[ngrp] = n;
    }
    T::sync();
    // scatter into stored order (all reads of the caller-order copy of vdw are done)
    for (int i = T::tid(); i < n; i += T::SIZE) {
        int pos = sh.inv[i];
        double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        sh.A.x[pos] = x; sh.A.y[pos] = y; sh.A.z[pos] = z;
        sh.A.xx[pos] = sq3(x, y, z);
        sh.vdw[pos] = vdw[i];
        sh.perm[pos] = i;
    }
    T::sync();
}

// sum_{i<n} term(i), strictly left to right (numpy's axis-0 reduction over rows), with the terms
// of eight rows fetched before they are added: the additions stay one dependent chain, the LDS
// reads (two levels through the permutation) no longer sit on it
// (B terms requested together, then added one after the other)
template <int B, class F>
PW_HD inline double seq_sum_blocks(int n, F term) {
    double s = term(0);
    int i = 1;
    for (; i + B <= n; i += B) {
        double v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) v[j] = term(i + j);
#pragma unroll
        for (int j = 0; j < B; ++j) s = s + v[j];
    }
    for (; i < n; ++i) s = s + term(i);
    return s;
}
template <class F>
PW_HD inline double seq_sum_blocked(int n, F term) { return seq_sum_blocks<8>(n, term); }

// Room for three columns of n terms in the caller's atom order, in the team memory that is idle while a frame is being
// prepared (window frames / optimiser states: UnitShared::scratch): the row-sequential sums of a unit -- centre of mass,
// centroid -- are ONE dependent chain of n additions each, and their terms used to be fetched by the summing lane
// itself, two reads deep through the permutation (4 us of a team with three lanes at work).  Written here by the whole
// team, the columns are read back contiguously, sixteen terms in flight.  nullptr: no room (or a one-thread team).
template <class T>
PW_HD inline ldouble* seq_sum_columns(const UnitShared& sh, int n) {
    const size_t np = (size_t)((n + 1) & ~1);
    if (T::SIZE < 64 || sh.scratch_bytes < 3 * np * 8) return nullptr;
    return (ldouble*)sh.scratch;
}
// the three sums by the first three threads (after a barrier behind the columns' writes); fin(c, sum)
template <class T, class FIN>
PW_HD inline void seq_sum_columns_add(const ldouble* cols, int n, FIN fin) {
    const size_t np = (size_t)((n + 1) & ~1);
    if (T::tid() < 3) {
        const ldouble* a = cols + (size_t)T::tid() * np;
        fin(T::tid(), seq_sum_blocks<16>(n, [&](int i) { return a[i]; }));
    }
}

// shifted copy S = A - c (elementwise), with |r|^2 and the row-sequential centroid
template <class T>
PW_HD inline void make_shifted(UnitShared& sh, int n, double cx, double cy, double cz) {
    ldouble* cols = seq_sum_columns<T>(sh, n);
    const size_t np = (size_t)((n + 1) & ~1);
    for (int i = T::tid(); i < n; i += T::SIZE) {
        double x = sh.A.x[i] - cx, y = sh.A.y[i] - cy, z = sh.A.z[i] - cz;
        sh.S.x[i] = x; sh.S.y[i] = y; sh.S.z[i] = z;
        sh.S.xx[i] = sq3(x, y, z);
        if (cols) { const int ci = sh.perm[i]; cols[ci] = x; cols[np + ci] = y; cols[2 * np + ci] = z; }
    }
    T::sync();
    // centroid: np.sum(coordinates, axis=0) / N -- rows added in the caller's atom order
    if (cols) {
        seq_sum_columns_add<T>(cols, n, [&](int c, double s) { sh.v->centroid[c] = s / (double)n; });
        T::sync();
        return;
    }
    auto cen_comp = [&](int c) {
        const ldouble* a = c == 0 ? sh.S.x : (c == 1 ? sh.S.y : sh.S.z);
        double s = seq_sum_blocked(n, [&](int i) { return a[sh.inv[i]]; });
        sh.v->centroid[c] = s / (double)n;
    };
    if (T::SIZE >= 3) {
        if (T::tid() < 3) cen_comp(T::tid());
    } else if (T::tid() == 0) {
        cen_comp(0); cen_comp(1); cen_comp(2);
    }
    T::sync();
}

// The N x N call of the distance primitive (utilities.py:366, X is Y) goes through OpenBLAS's dsyrk, whose
// SkylakeX kernels do NOT sum the three products of an entry in one order everywhere.  When N % 8 >= 4 the four
// atoms [8*(N/8), 8*(N/8)+4) are an edge tile of the last row panel, and an entry between one of them and atom c is
//   fma(z,z', x*x' + y*y')        when c sits in the first 12*floor(w/12) columns of the kernel call that covers it
//                                 (the 4x12 micro-kernel),
//   fma(z,z', fma(y,y', x*x'))    otherwise -- the order of every other entry of the matrix.
// The kernel calls of the last row panel: ONE for all columns left of the panel (w = where the panel starts), then
// one per 32 columns of the panel itself (w = min(32, N - 32*floor(c/32))).  The panel starts at 0 up to N = 192
// (GEMM_P) and at 32*ceil(floor(N/2)/32) from there to 382; from 383 atoms the BLAS shares the product among its
// threads and the panels depend on how many the machine has -- the reference's last bit does too, and nothing is
// restated there.  Established entry by entry against numpy's X @ X.T (tests/tools/distance_order_probe.py --rule),
// against sklearn on the molecules that exposed it (tests/golden/edge_tile.npz), against the reference's max_dim on
// tens of thousands of random molecules.  DESIGN.md section 7.
struct GramEdgeRule {
    int t0, panel, panel_wide, nfull, limlast;
    PW_HD static bool applies(int n) { return n % 8 >= 4; }
    // where the last row panel of the BLAS's level-3 driver starts (GEMM_P = 192, panels rounded to 32): a panel of 192
    // rows while at least 384 are left, then what is left in two halves, then the rest.  This is the recurrence of ONE
    // BLAS thread (probed entry by entry up to 8197 atoms, round 5); up to 382 atoms it is what every thread count
    // gives, from 383 OpenBLAS shares the product among its threads and the reference's own last bit follows the core
    // count -- the platform restated here is OPENBLAS_NUM_THREADS=1.
    PW_HD static int last_panel_start(int n) {
        int start = 0;
        for (;;) {
            const int rem = n - start;
            const int mi = rem >= 384 ? 192 : (rem > 192 ? 32 * ((rem / 2 + 31) / 32) : rem);
            if (start + mi >= n) return start;
            start += mi;
        }
    }
    PW_HD explicit GramEdgeRule(int n) {
        t0 = applies(n) ? 8 * (n / 8) : 0x40000000;
        panel = last_panel_start(n);
        panel_wide = 12 * (panel / 12);
        nfull = 32 * (n / 32);
        limlast = 12 * ((n - nfull) / 12);
    }
    PW_HD bool is_edge(int o) const { return (unsigned)(o - t0) < 4u; }
    PW_HD bool in_wide_kernel(int o) const {
        return o < panel ? o < panel_wide : (o & 31) < (o < nfull ? 24 : limlast);
    }
    // oi, oj: the caller's numbering of the two atoms
    PW_HD bool pair_uses_edge_order(int oi, int oj) const {
        return (is_edge(oi) && in_wide_kernel(oj)) || (is_edge(oj) && in_wide_kernel(oi));
    }
};
template <bool EDGE>
PW_HD inline __attribute__((always_inline)) double pw_gram_nn(double xi, double yi, double zi, double xj, double yj,
                                                              double zj, int oi, int oj, const GramEdgeRule& er) {
    if (EDGE && er.pair_uses_edge_order(oi, oj)) return pw_fma(zi, zj, xi * xj + yi * yj);
    return pw_fma(zi, zj, pw_fma(yi, yj, xi * xj));
}

// ---- max_dim over the few atoms that can be an end of it ------------------------------------------------------
// max_dim is the one N x N block of the path: the largest d_ij + (vdw_i + vdw_j) over all pairs.  Only atoms of the
// outermost shell can be an end of it.  With s_i = |a_i - c| + vdw_i about any centre c, every entry is at most
// s_i + s_j (triangle inequality), and the best partner of the atom with the largest s -- one row of the matrix, with
// the reference's arithmetic -- is a lower bound L of the maximum: an atom with s_i + s_max < L cannot be in a maximal
// pair.  For a cage that leaves a dozen atoms of 168, and the N x N block becomes a hundred pairs, each evaluated
// with the arithmetic of the full block (Gram form, edge rule, row norm first) -- the same maximum, the same first
// index.  The margin covers what the Gram form's computed distance can exceed the true one by (its cancellation
// error grows with the square of the coordinates: 4e-15 * max |a|^2 / L) a million times over for a molecule near the
// origin.  More than 64 atoms left (a hollow shell: every atom has an antipode) or a one-lane team: false, nothing
// done -- the caller runs the full block.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T, bool VALUE_ONLY, bool EDGE>
__device__ inline __attribute__((always_inline)) bool team_max_dim_few(UnitShared& sh, const Frame& F, int n,
                                                                       const GramEdgeRule& er, const double* centre) {
    auto& v = *sh.v;
    const double cx = centre ? centre[0] : 0.0, cy = centre ? centre[1] : 0.0, cz = centre ? centre[2] : 0.0;
    // 1. s_i, its maximum (any atom that has it), the largest squared norm
    double smax = -PW_INF, nmax = 0.0;
    int imax = 0x7fffffff;
    for (int i = T::tid(); i < n; i += T::SIZE) {
        const double dx = F.x[i] - cx, dy = F.y[i] - cy, dz = F.z[i] - cz;
        const double si = pw_sqrt(dx * dx + dy * dy + dz * dz) + F.vdw[i];
        if (si > smax) { smax = si; imax = i; }
        nmax = pw_max(nmax, F.xx[i]);
    }
    T::wave_argmax(smax, imax);
    nmax = -T::wave_min(-nmax);
    if (T::lane() == 0) { v.red_v[T::wave()] = smax; v.red_i[T::wave()] = imax; v.red_v[8 + T::wave()] = nmax; }
    T::sync();
    {
        double bb = v.red_v[0], nn = v.red_v[8];
        int bi = v.red_i[0];
        for (int w = 1; w < T::NWAVES; ++w) {
            if (v.red_v[w] > bb || (v.red_v[w] == bb && v.red_i[w] < bi)) { bb = v.red_v[w]; bi = v.red_i[w]; }
            nn = pw_max(nn, v.red_v[8 + w]);
        }
        smax = bb; imax = bi; nmax = nn;
    }
    T::sync();
    if (!(smax < PW_INF) || imax >= n) return false;
    // 2. L: the best partner of that atom, the reference's arithmetic (the diagonal entry included)
    const double xi = F.x[imax], yi = F.y[imax], zi = F.z[imax], xxi = F.xx[imax], vi = F.vdw[imax];
    const int oi = F.perm[imax];
    double L = -PW_INF;
    for (int j = T::tid(); j < n; j += T::SIZE) {
        const int oj = F.perm[j];
        double d = 0.0;
        if (j != imax) {
            const double gg = pw_gram_nn<EDGE>(xi, yi, zi, F.x[j], F.y[j], F.z[j], oi, oj, er);
            const double d2 = (oi < oj) ? pw_m2add(gg, xxi) + F.xx[j] : pw_m2add(gg, F.xx[j]) + xxi;
            d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
        }
        L = pw_max(L, d + (vi + F.vdw[j]));
    }
    L = -T::wave_min(-L);
    if (T::lane() == 0) v.red_v[T::wave()] = L;
    T::sync();
    L = v.red_v[0];
    for (int w = 1; w < T::NWAVES; ++w) L = pw_max(L, v.red_v[w]);
    T::sync();
    if (!(L < PW_INF)) return false;
    // 3. the atoms that can reach it, listed (any order: the pairs below are compared by the caller's numbering)
    const double margin = 1e-6 + 4e-15 * nmax / pw_max(L, 1e-3);
    const double thr = (L - smax) - margin;
    if (T::tid() == 0) v.md_count = 0;
    T::sync();
    for (int i0 = 0; i0 < n; i0 += T::SIZE) {
        const int i = i0 + T::tid();
        bool in = false;
        if (i < n) {
            const double dx = F.x[i] - cx, dy = F.y[i] - cy, dz = F.z[i] - cz;
            in = pw_sqrt(dx * dx + dy * dy + dz * dz) + F.vdw[i] >= thr;
        }
        const unsigned long long mk = T::ballot(in);
        int base = 0;
        if (T::lane() == 0 && mk) base = atomicAdd((PW_LDS int*)&v.md_count, (int)__builtin_popcountll(mk));
        base = __builtin_amdgcn_readfirstlane(base);
        const int at = base + (int)__builtin_popcountll(mk & ((1ull << T::lane()) - 1ull));
        if (in && at < 64) v.md_list[at] = i;
    }
    T::sync();
    const int m = v.md_count;
    if (m > 64) { T::sync(); return false; }
    // 4. every pair of the list (row < column in the caller's numbering; the diagonal with it)
    double bv = -PW_INF, bidx = PW_INF;
    for (int p = T::tid(); p < m * m; p += T::SIZE) {
        const int a = p / m, b = p - a * m;
        const int ia = v.md_list[a], ib = v.md_list[b];
        const int oa = F.perm[ia], ob = F.perm[ib];
        if (oa > ob) continue;
        double d = 0.0;
        if (ia != ib) {
            const double gg = pw_gram_nn<EDGE>(F.x[ia], F.y[ia], F.z[ia], F.x[ib], F.y[ib], F.z[ib], oa, ob, er);
            const double d2 = pw_m2add(gg, F.xx[ia]) + F.xx[ib];              // (oa < ob: the row norm first)
            d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
        }
        const double val = d + (F.vdw[ia] + F.vdw[ib]);
        const double idx = (double)(oa * n + ob);
        if (val > bv || (val == bv && idx < bidx)) { bv = val; bidx = idx; }
    }
    int ii = (int)(bidx < 2147483647.0 ? bidx : 2147483647.0);
    T::wave_argmax(bv, ii);
    if (T::lane() == 0) { v.red_v[T::wave()] = bv; v.red_i[T::wave()] = ii; }
    T::sync();
    if (T::tid() == 0) {
        double bb = v.red_v[0];
        int bi = v.red_i[0];
        for (int w = 1; w < T::NWAVES; ++w) {
            const double vv = v.red_v[w];
            const int vi2 = v.red_i[w];
            if (vv > bb || (vv == bb && vi2 < bi)) { bb = vv; bi = vi2; }
        }
        v.maxd = bb;
        if (!VALUE_ONLY) { v.maxd_i = bi / n; v.maxd_j = bi % n; }
    }
    T::sync();
    return true;
}
#endif

// max_dim over frame F (utilities.py:355-372); result in sh.v->maxd*, all threads.  VALUE_ONLY: the
// callers that only want the diameter of the shifted molecule (the radius of the sampling sphere)
// skip the second pass; maxd_i / maxd_j are then not touched.
template <class T, bool VALUE_ONLY = false, bool EDGE = false>
PW_HD inline __attribute__((always_inline)) void team_max_dim_body(UnitShared& sh, const Frame& F, int n,
                                                                   double* item_best = nullptr, const double* centre = nullptr) {
    const GramEdgeRule er(n);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_MAXDIM_FEW)
    // (the atoms of the outermost shell only, when they are few: team_max_dim_few)
    if (T::WSIZE == 64 && team_max_dim_few<T, VALUE_ONLY, EDGE>(sh, F, n, er, centre)) return;
#else
    (void)centre;
#endif
    // max over pairs (diagonal included) of d_ij + (vdw_i + vdw_j), first maximum in row-major
    // order of the caller's numbering (utilities.py:355-372).  Two passes over the upper triangle:
    //   1. value only -- inside one radius group of the column the maximum of the sum is at the
    //      maximum squared distance (sqrt and the addition of a common constant are monotone), so
    //      the column loop carries squared distances and takes one sqrt per (row, group);
    //   2. the winner's index: only pairs whose squared distance is within a few ulps of what
    //      the maximum requires are evaluated exactly and compared by index.
    // Work items: row p is folded with row n-1-p (together they have n-1 columns above the diagonal,
    // whatever p is), and the columns of a folded row are dealt to K threads by their residue mod K,
    // K varying fastest over the threads: every item has (n-1)/K columns, the threads of a wave walk
    // the radius groups in step (their counts per group differ by one at most) and read K adjacent
    // columns at a time.  (Half rows per thread, the round-1 scheme, left the triangle's short rows
    // idle: 164 column steps per thread for CC3 against 63 here.)
    const auto& C = *F.cls;
    const int ngrp = C.k;
    const int R = (n + 1) / 2;
    int K = 1;
    {
        long best_cost = 0x7fffffffffffffffl;
        for (int k = 1; k <= 4; ++k) {
            long passes = ((long)R * k + T::SIZE - 1) / T::SIZE;
            long cost = passes * ((n - 1 + 64 / k + k - 1) / k);
            if (cost < best_cost) { best_cost = cost; K = k; }
        }
    }
    const int nitem = R * K;
    double best = -PW_INF;
    if (ngrp > 0) {
        for (int t = T::tid(); t < nitem; t += T::SIZE) {
            const int p = t / K, q = t - p * K;
            double ibest = -PW_INF;                                     // this item's maximum
            for (int side = 0; side < 2; ++side) {
                const int i = side == 0 ? p : n - 1 - p;
                if (side == 1 && i == p) break;                         // (the middle row of an odd n)
                const double xi = F.x[i], yi = F.y[i], zi = F.z[i], xxi = F.xx[i], vi = F.vdw[i];
                const int oi = F.perm[i];
                if (q == 0) ibest = pw_max(ibest, 0.0 + (vi + vi));     // the diagonal entry
                for (int g = 0; g < ngrp; ++g) {
                    int lo = C.off[g] > i + 1 ? C.off[g] : i + 1;
                    const int hi = C.off[g + 1];
                    lo += (q - lo % K + K) % K;                         // first column of residue q
                    if (lo >= hi) continue;
                    double m2 = -PW_INF;
                    int j = lo;
                    // entry (row, column) of the reference's matrix has row < column in the caller's
                    // numbering: the row norm is added first
                    for (; j + 7 * K < hi; j += 8 * K) {
                        double ax[8], ay[8], az[8], aq[8];
                        int ap[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int c = j + u * K;
                            ax[u] = F.x[c]; ay[u] = F.y[c]; az[u] = F.z[c]; aq[u] = F.xx[c]; ap[u] = F.perm[c];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            double gg = pw_gram_nn<EDGE>(xi, yi, zi, ax[u], ay[u], az[u], oi, ap[u], er);
                            double d2 = (oi < ap[u]) ? pw_m2add(gg, xxi) + aq[u] : pw_m2add(gg, aq[u]) + xxi;
                            m2 = __builtin_fmax(m2, d2);
                        }
                    }
                    for (; j < hi; j += K) {
                        double gg = pw_gram_nn<EDGE>(xi, yi, zi, F.x[j], F.y[j], F.z[j], oi, F.perm[j], er);
                        double d2 = (oi < F.perm[j]) ? pw_m2add(gg, xxi) + F.xx[j] : pw_m2add(gg, F.xx[j]) + xxi;
                        m2 = __builtin_fmax(m2, d2);
                    }
                    double d = pw_sqrt(m2 > 0.0 ? m2 : 0.0);
                    ibest = pw_max(ibest, d + (vi + C.vdw[g]));
                }
            }
            best = pw_max(best, ibest);
            if (!VALUE_ONLY && item_best) item_best[t] = ibest;
        }
    }
    // team maximum of the values
    best = -T::wave_min(-best);
    if (T::lane() == 0) sh.v->red_v[T::wave()] = best;
    T::sync();
    double vmax = sh.v->red_v[0];
    for (int w = 1; w < T::NWAVES; ++w) vmax = pw_max(vmax, sh.v->red_v[w]);
    T::sync();
    if (VALUE_ONLY && ngrp > 0) {
        if (T::tid() == 0) sh.v->maxd = vmax;
        T::sync();
        return;
    }
    // pass 2: smallest row-major index among the pairs that reach vmax
    double bidx = PW_INF;
    if (ngrp > 0) {
        for (int t = T::tid(); t < nitem; t += T::SIZE) {
            if (item_best && item_best[t] != vmax) continue;            // (written by this very thread)
            const int p = t / K, q = t - p * K;
            for (int side = 0; side < 2; ++side) {
                const int i = side == 0 ? p : n - 1 - p;
                if (side == 1 && i == p) break;
                const double xi = F.x[i], yi = F.y[i], zi = F.z[i], xxi = F.xx[i], vi = F.vdw[i];
                const int oi = F.perm[i];
                if (q == 0 && 0.0 + (vi + vi) == vmax) bidx = pw_min(bidx, (double)(oi * n + oi));
                for (int g = 0; g < ngrp; ++g) {
                    int lo = C.off[g] > i + 1 ? C.off[g] : i + 1;
                    const int hi = C.off[g + 1];
                    lo += (q - lo % K + K) % K;
                    if (lo >= hi) continue;
                    const double c = vi + C.vdw[g];
                    // d + c == vmax needs d >= vmax - c - ulp(vmax); squared, with margin
                    const double need = (vmax - c) - 1e-15 * vmax;
                    const double thr = need > 0.0 ? need * need * (1.0 - 1e-15) : -PW_INF;
                    auto candidate = [&](double d2, int oj) {
                        double d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
                        if (d + c == vmax) {
                            int idx = oi < oj ? oi * n + oj : oj * n + oi;
                            bidx = pw_min(bidx, (double)idx);
                        }
                    };
                    int j = lo;
                    for (; j + 7 * K < hi; j += 8 * K) {
                        double ax[8], ay[8], az[8], aq[8];
                        int ap[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int cc = j + u * K;
                            ax[u] = F.x[cc]; ay[u] = F.y[cc]; az[u] = F.z[cc]; aq[u] = F.xx[cc]; ap[u] = F.perm[cc];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            double gg = pw_gram_nn<EDGE>(xi, yi, zi, ax[u], ay[u], az[u], oi, ap[u], er);
                            double d2 = (oi < ap[u]) ? pw_m2add(gg, xxi) + aq[u] : pw_m2add(gg, aq[u]) + xxi;
                            if (d2 >= thr) candidate(d2, ap[u]);
                        }
                    }
                    for (; j < hi; j += K) {
                        int oj = F.perm[j];
                        double gg = pw_gram_nn<EDGE>(xi, yi, zi, F.x[j], F.y[j], F.z[j], oi, oj, er);
                        double d2 = (oi < oj) ? pw_m2add(gg, xxi) + F.xx[j] : pw_m2add(gg, F.xx[j]) + xxi;
                        if (d2 >= thr) candidate(d2, oj);
                    }
                }
            }
        }
    } else {
        // more radii than groups: plain scan (value and index together)
        double bv = -PW_INF;
        for (int r = T::tid(); r < n; r += T::SIZE) {
            int i = (r & 1) ? (n - 1 - (r >> 1)) : (r >> 1);
            double xi = F.x[i], yi = F.y[i], zi = F.z[i], xxi = F.xx[i], vi = F.vdw[i];
            int oi = F.perm[i];
            for (int j = i; j < n; ++j) {
                double d = 0.0;
                int oj = F.perm[j];
                if (j != i) {
                    double gg = pw_gram_nn<EDGE>(xi, yi, zi, F.x[j], F.y[j], F.z[j], oi, oj, er);
                    double d2 = (oi < oj) ? pw_m2add(gg, xxi) + F.xx[j] : pw_m2add(gg, F.xx[j]) + xxi;
                    d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
                }
                double v = d + (vi + F.vdw[j]);
                double idx = (double)(oi < oj ? oi * n + oj : oj * n + oi);
                if (v > bv || (v == bv && idx < bidx)) { bv = v; bidx = idx; }
            }
        }
        // two-level reduction on (value, index)
        int ii = (int)(bidx < 2147483647.0 ? bidx : 2147483647.0);
        T::wave_argmax(bv, ii);
        if (T::lane() == 0) { sh.v->red_v[T::wave()] = bv; sh.v->red_i[T::wave()] = ii; }
        T::sync();
        if (T::tid() == 0) {
            double bb = sh.v->red_v[0];
            int bi = sh.v->red_i[0];
            for (int w = 1; w < T::NWAVES; ++w) {
                double v = sh.v->red_v[w];
                int vi2 = sh.v->red_i[w];
                if (v > bb || (v == bb && vi2 < bi)) { bb = v; bi = vi2; }
            }
            sh.v->maxd = bb;
            sh.v->maxd_i = bi / n;
            sh.v->maxd_j = bi % n;
        }
        T::sync();
        return;
    }
    bidx = T::wave_min(bidx);
    if (T::lane() == 0) sh.v->red_v[T::wave()] = bidx;
    T::sync();
    if (T::tid() == 0) {
        double bi = sh.v->red_v[0];
        for (int w = 1; w < T::NWAVES; ++w) bi = pw_min(bi, sh.v->red_v[w]);
        int idx = (int)bi;
        sh.v->maxd = vmax;
        sh.v->maxd_i = idx / n;
        sh.v->maxd_j = idx % n;
    }
    T::sync();
}

// (the edge order is its own instantiation: a molecule without an edge tile -- N % 8 < 4, the CC3 cage -- runs the
// loops as they were)
template <class T, bool VALUE_ONLY = false>
PW_HD inline __attribute__((always_inline)) void team_max_dim_impl(UnitShared& sh, const Frame& F, int n,
                                                                   double* item_best = nullptr, const double* centre = nullptr) {
    if (GramEdgeRule::applies(n)) team_max_dim_body<T, VALUE_ONLY, true>(sh, F, n, item_best, centre);
    else team_max_dim_body<T, VALUE_ONLY, false>(sh, F, n, item_best, centre);
}

// centre: a point near the middle of the molecule (the centre of mass of the input frame; null: the origin, where a
// shifted frame has it) -- only the bounds of team_max_dim_few use it, never a result
template <class T, bool VALUE_ONLY = false>
PW_NOINLINE PW_HD inline void team_max_dim(UnitShared& sh, const Frame& F, int n, double* item_best = nullptr,
                                           const double* centre = nullptr) {
    PW_ASSUME_TEAM_SH(sh);
    team_max_dim_impl<T, VALUE_ONLY>(sh, F, n, item_best, centre);
}

// ---- stage: basic -------------------------------------------------------------------------
template <class T>
PW_HD inline __attribute__((always_inline)) void stage_basic_impl(UnitShared& sh, TeamWorkspace* ws, int n,
                                                                  pw_unit_out* out, bool com_only) {
    auto& v = *sh.v;
    (void)ws;
    // centre of mass: per component the row-sequential sum of x_i*m_i over the mass
    ldouble* cols = seq_sum_columns<T>(sh, n);
    if (cols) {
        const size_t np = (size_t)((n + 1) & ~1);
        for (int i = T::tid(); i < n; i += T::SIZE) {
            const int pos = sh.inv[i];
            const double m = sh.mass[i];
            cols[i] = sh.A.x[pos] * m; cols[np + i] = sh.A.y[pos] * m; cols[2 * np + i] = sh.A.z[pos] * m;
        }
    }
    if (T::tid() == 0 && !v.mw_given) v.mw = molecular_weight_sum((const double*)sh.mass, n);
    T::sync();
    if (cols) {
        seq_sum_columns_add<T>(cols, n, [&](int c, double s) { v.com[c] = s / v.mw; });
    } else {
    auto com_comp = [&](int c) {
        const ldouble* a = c == 0 ? sh.A.x : (c == 1 ? sh.A.y : sh.A.z);
        double s = seq_sum_blocked(n, [&](int i) { return a[sh.inv[i]] * sh.mass[i]; });
        v.com[c] = s / v.mw;
    };
    if (T::SIZE >= 3) {
        if (T::tid() < 3) com_comp(T::tid());
    } else if (T::tid() == 0) {
        com_comp(0); com_comp(1); com_comp(2);
    }
    }
    T::sync();
    if (com_only) return;
    {
        const double com3[3] = {v.com[0], v.com[1], v.com[2]};
        team_max_dim<T>(sh, sh.A, n, 2 * n + 2 <= ws->p_cap ? ws->vals : nullptr, com3);
    }
    if (T::wave() == 0) {
        int arg;
        double g = wave_gap<T>(sh.A, n, v.com[0], v.com[1], v.com[2], &arg);
        if (T::lane() == 0) { v.pore_g = g; v.pore_atom = arg; v.n_eval += 1; }
    }
    T::sync();
    if (T::tid() == 0) {
        out->n_atoms = n;
        out->mw = v.mw;
        out->com[0] = v.com[0]; out->com[1] = v.com[1]; out->com[2] = v.com[2];
        out->maxd = v.maxd; out->maxd_i = v.maxd_i; out->maxd_j = v.maxd_j;
        out->pore_d = v.pore_g * 2.0;
        out->pore_atom = v.pore_atom;
        double r = out->pore_d / 2.0;
        out->pore_vol = FOUR_THIRDS_PI * pw_cube_np(r);
    }
    T::sync();
}

template <class T>
PW_NOINLINE PW_HD inline void stage_basic(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                           bool com_only) {
    PW_ASSUME_TEAM_SH(sh);
    stage_basic_impl<T>(sh, ws, n, out, com_only);
}

// forward-difference gradient step exactly as scipy.optimize._numdiff (2-point,
// abs_step 1e-8, _adjust_scheme_to_bounds '1-sided')
PW_HD inline double fd_step(double x, double lb, double ub) {
    double h = 1e-8;
    // "cannot have a zero step": for huge |x| fall back to the relative step sqrt(eps) * sign * max(1, |x|)
    if ((x + h) - x == 0.0) h = 1.4901161193847656e-08 * (x >= 0.0 ? 1.0 : -1.0) * pw_max(1.0, pw_abs(x));
    double lower = x - lb, upper = ub - x;
    double xh = x + h;
    bool violated = (xh < lb) || (xh > ub);
    bool fitting = pw_abs(h) <= pw_max(lower, upper);
    if (violated && fitting) h = -h;
    else if (!fitting) h = (upper >= lower) ? upper : -lower;
    return h;
}

// The objective of opt_pore_diameter as scipy.optimize hands it to L-BFGS-B: f = -pore_diameter(x) and its
// forward-difference gradient (fd_step), the four points evaluated at once; scipy's ScalarFunction re-uses f and g
// when asked for the point it evaluated last (_differentiable_functions.py: fun_and_grad).  nfev counts the
// evaluations as scipy does (what its driver compares with maxfun).
template <class T>
struct PoreObjective {
    const Frame& A;
    int n;
    const double* lo;
    const double* up;
    PW_LDS int* cand;
    bool have_last;
    double lx, ly, lz, lf, lg[3];
    int nfev;
    unsigned long long* prof;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
    NearGap4<T> near_gap;
#endif
    PW_HD PoreObjective(const Frame& A_, int n_, const double* lo_, const double* up_, PW_LDS int* cand_)
        : A(A_), n(n_), lo(lo_), up(up_), cand(cand_), have_last(false), lx(0.0), ly(0.0), lz(0.0), lf(0.0), nfev(0),
          prof(nullptr) {
        lg[0] = lg[1] = lg[2] = 0.0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
        near_gap.init();
#endif
    }
    PW_HD __attribute__((always_inline)) void operator()(const double* xq, double& fo, double* go) {
        const double px = xq[0], py = xq[1], pz = xq[2];
        if (!(have_last && px == lx && py == ly && pz == lz)) {
            LB_F0(f_pre);
            double qx[4] = {px, px, px, px}, qy[4] = {py, py, py, py}, qz[4] = {pz, pz, pz, pz};
            double dxs[3];
            // the usual case first, all three coordinates with one branch: the step 1e-8 is representable at x, x + h
            // stays inside the box and fits it -- fd_step then returns h = 1e-8 as it is
            bool plain = true;
            double x1s[3];
            for (int c = 0; c < 3; ++c) {
                const double xc = c == 0 ? px : (c == 1 ? py : pz);
                const double xh = xc + 1e-8;
                x1s[c] = xh;
                dxs[c] = xh - xc;
                plain = plain && (dxs[c] != 0.0) && !(xh < lo[c]) && !(xh > up[c]) &&
                        (1e-8 <= pw_max(xc - lo[c], up[c] - xc));
            }
            if (!plain) {
                for (int c = 0; c < 3; ++c) {
                    double xc = c == 0 ? px : (c == 1 ? py : pz);
                    double h = fd_step(xc, lo[c], up[c]);
                    double x1 = xc + h;
                    dxs[c] = x1 - xc;
                    x1s[c] = x1;
                }
            }
            qx[1] = x1s[0]; qy[2] = x1s[1]; qz[3] = x1s[2];
            double gv[4];
            LB_F1(27, f_pre);
#if defined(PW_PROFILE) && defined(PW_LB_FINE) && defined(__HIP_DEVICE_COMPILE__)
            long long t_g4 = clock64();
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
            if (T::WSIZE == 64) near_gap.eval(A, n, cand, qx, qy, qz, gv);
            else
#endif
            wave_gap4<T>(A, n, qx, qy, qz, gv);
#if defined(PW_PROFILE) && defined(PW_LB_FINE) && defined(__HIP_DEVICE_COMPILE__)
            if (prof && T::lane() == 0) atomicAdd(&prof[24], (unsigned long long)(clock64() - t_g4));
#endif
            LB_F0(f_post);
            double f0 = -(gv[0] * 2.0);
            for (int c = 0; c < 3; ++c) {
                double f1 = -(gv[c + 1] * 2.0);
                lg[c] = (f1 - f0) / dxs[c];
            }
            lf = f0;
            lx = px; ly = py; lz = pz;
            have_last = true;
            nfev += 4;
            LB_F1(28, f_post);
        }
        fo = lf;
        go[0] = lg[0]; go[1] = lg[1]; go[2] = lg[2];
    }
};

// ---- stage: optimised pore (wave 0) --------------------------------------------------------
template <class T>
PW_HD inline __attribute__((always_inline)) void stage_opt_impl(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                                                const pw_params& prm) {
    (void)ws;
    auto& v = *sh.v;
    if (T::wave() == 0) {
        const Frame A = sh.A;                // (a copy: `sh` itself is behind a generic pointer, re-read after every store)
        Lbfgsb<3> opt;                       // scalars in registers, arrays in LDS
        Lbfgsb<3>* S = &opt;
        LbMem<3>* Smem = (LbMem<3>*)sh.lb[0];
        PW_ASSUME_LDS(Smem);
        double r = v.pore_g;  // pore_diameter / 2
        double lo[3], up[3], x0[3];
        int nbd[3] = {2, 2, 2};
        for (int c = 0; c < 3; ++c) x0[c] = (prm.opt_flags & PW_OPT_CUSTOM_START) ? prm.opt_x0[c] : v.com[c];
        // user-supplied start: the default box is built around it (utilities.py:412-421)
        if ((prm.opt_flags & PW_OPT_CUSTOM_START) && !(prm.opt_flags & PW_OPT_CUSTOM_BOUNDS))
            r = wave_gap<T>(A, n, x0[0], x0[1], x0[2], nullptr);
        bool bad;
        if (prm.opt_flags & PW_OPT_CUSTOM_BOUNDS) {
            bad = false;
            for (int c = 0; c < 3; ++c) {
                lo[c] = prm.opt_lo[c];
                up[c] = prm.opt_hi[c];
                bool has_lo = lo[c] > -PW_INF, has_up = up[c] < PW_INF;
                nbd[c] = has_lo ? (has_up ? 2 : 1) : (has_up ? 3 : 0);
                if (lo[c] > up[c]) bad = true;
                // scipy clips the start into the box (_lbfgsb_py.py: x0 = np.clip(x0, lb, ub))
                x0[c] = x0[c] < lo[c] ? lo[c] : (x0[c] > up[c] ? up[c] : x0[c]);
            }
        } else {
            for (int c = 0; c < 3; ++c) {
                lo[c] = x0[c] - r;
                up[c] = x0[c] + r;
            }
            bad = !(r > 0.0);
        }
        int nit = 0, nfev = 0;
        if (!bad) {
            S->template setup<T>(Smem, x0, lo, up, nbd, 1e7, 1e-5, 20);
#if defined(PW_PROFILE) && !defined(PW_ZPROF)
            S->prof = PW_PROF_BASE(ws);
#endif
#if defined(PW_PROFILE) && defined(PW_LB_FINE) && !defined(PW_ZPROF)
            Smem->prof_fine = PW_PROF_BASE(ws);
#endif
            PoreObjective<T> fg(A, n, lo, up, (PW_LDS int*)Smem->cand);
#if defined(PW_PROFILE) && !defined(PW_ZPROF)
            fg.prof = PW_PROF_BASE(ws);
#endif
            // scipy's driver: maxiter = maxfun = 15000, both tested at a new iterate only
            // (_lbfgsb_py.py: "interruptions due to maxfun are postponed")
            PW_T0(t_s);
            S->template minimize<T>(fg, 15000, 15000, &nit);
            T::wave_sync();
            PW_T1(ws, 0, t_s);
            nfev = fg.nfev;
        }
        double cx = v.com[0], cy = v.com[1], cz = v.com[2];
        if (!bad) { cx = S->x[0]; cy = S->x[1]; cz = S->x[2]; }
        int arg;
        double g = wave_gap<T>(A, n, cx, cy, cz, &arg);
        if (T::lane() == 0) {
            v.opt_c[0] = cx; v.opt_c[1] = cy; v.opt_c[2] = cz;
            v.opt_g = g;
            v.opt_atom = arg;
            v.n_eval += nfev + 1;
            if (bad) v.status |= PW_ST_NEGATIVE_PORE;
            out->pore_opt_d = g * 2.0;
            out->pore_opt_atom = arg;
            out->pore_opt_c[0] = cx; out->pore_opt_c[1] = cy; out->pore_opt_c[2] = cz;
            double rr = out->pore_opt_d / 2.0;
            out->pore_vol_opt = FOUR_THIRDS_PI * pw_cube_np(rr);
            out->opt_nit = nit;
            out->opt_nfev = nfev;
            out->opt_task = bad ? -1 : S->task;
            out->opt_msg = bad ? 0 : S->msg;
        }
    }
    T::sync();
}

template <class T>
PW_NOINLINE PW_HD inline void stage_opt(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                        const pw_params& prm) {
    PW_ASSUME_TEAM_STATE(sh, prm);
    stage_opt_impl<T>(sh, ws, n, out, prm);
}

// ---- stage: average diameter ---------------------------------------------------------------
// INL: everything inlined into the caller (the average-diameter launch has a kernel of its own,
// whose register budget then covers the whole stage: three waves per SIMD instead of two)
template <class T, bool INL>
PW_HD inline __attribute__((always_inline)) void stage_average_impl(UnitShared& sh, TeamWorkspace* ws, int n,
                                                                    pw_unit_out* out, const pw_params& prm) {
    auto& v = *sh.v;
    PW_T0(t_a0);
    make_shifted<T>(sh, n, v.com[0], v.com[1], v.com[2]);
    // preserve the input-frame max_dim: the shifted frame's replaces it only here
    double keep_d = v.maxd;
    int keep_i = v.maxd_i, keep_j = v.maxd_j;
    T::sync();
    {
        double* ib = 2 * n + 2 <= ws->p_cap ? ws->vals : nullptr;     // per-item maxima (free until the rays)
        if (INL) team_max_dim_impl<T, true>(sh, sh.S, n, ib); else team_max_dim<T, true>(sh, sh.S, n, ib);
    }
    double radius = v.maxd;
    T::sync();
    if (T::tid() == 0) { v.maxd = keep_d; v.maxd_i = keep_i; v.maxd_j = keep_j; }
    int P = sampling_count(radius, prm.adjust_average);
    if (P > ws->p_cap) {
        // more rays than this launch's workspace holds: flagged, and the value is NOT a number (the host
        // sizes the workspace from the adjust knob; pw_analysis_batch re-runs with the capacity asked for)
        if (T::tid() == 0) { v.status |= PW_ST_POINTS_OVERFLOW; out->avg_d = __builtin_nan(""); out->n_points_avg = P; }
        T::sync();
        return;
    }
    Sphere sp;
    sp.init(radius, P);
    ScratchArena arena;
    arena.init(sh);
    double* vals = (double*)arena.take((size_t)P * 8);
    if (!vals) vals = ws->vals;
    double* packed = vals;     // the hits are compacted in place (a slot never moves to a later one)
    unsigned char* flag = (unsigned char*)arena.take((size_t)P);
    if (!flag) flag = ws->flag;
    int* s_tab = (int*)arena.take(324 * 4);
    // np_sum_team: eight accumulators per 64-element slot of (at most) P values, and never less
    // than the 128 words its tree walk borrows
    size_t acc_words = 8 * (size_t)(((P < 8192 ? P : 8192) + 63) / 64);
    if (acc_words < 128) acc_words = 128;
    double* s_acc = (double*)arena.take(acc_words * 8);
    double* s_leaf = (double*)arena.take(256 * 8);
    if (!s_tab || !s_acc || !s_leaf) { s_tab = ws->leaf_tab; s_acc = ws->acc8; s_leaf = ws->leaf; }
    double cen[3] = {v.centroid[0], v.centroid[1], v.centroid[2]};
    if (T::wave() == 0) PW_T1(ws, 27, t_a0);
    PW_T0(t_a1);
    // atom by atom over the rays inside each atom's cone (team_ray_tests) when the ray vectors fit the
    // launch's LDS scratch; the dense scan (every ray against every atom) otherwise
    bool done = false;
    {
        ScratchArena a2 = arena;
        double* apts = (double*)a2.take((size_t)P * 24);
        if (apts) {
            team_sphere_points<T>(ws, sp, [&](int k, double x, double y, double z) {
                apts[k] = x; apts[P + k] = y; apts[2 * P + k] = z;
                vals[k] = -1.0;
                flag[k] = 0;
            });
            ConeBand* bands = (ConeBand*)a2.take((size_t)n * sizeof(ConeBand));
            int cap = (int)(a2.left / 4);
            unsigned* pairs = (unsigned*)a2.take((size_t)cap * 4);
            if (cap < 4 * P) { pairs = (unsigned*)ws->knn; cap = 16 * ws->p_cap; }
            if (!bands && (size_t)n * sizeof(ConeBand) <= (size_t)ws->p_cap * 16) bands = (ConeBand*)(ws->knn + 8 * (size_t)ws->p_cap);
            // (apts comes from the arena: team memory)
            PlanarRays<decltype(PW_AS_LDS(apts))> getp{PW_AS_LDS(apts), P};
            if (bands)
                done = team_ray_tests<T, true>(sh.S, n, cen, sp, getp, bands, pairs, cap, (PW_LDS int*)&v.red_i[0], flag, 1, vals);
        }
    }
    if (done) {
    } else if (T::SIZE > 1) {
        // four rays per thread and pass over the atoms
        constexpr int NR = 4;
        for (int k0 = T::tid(); k0 < P; k0 += NR * T::SIZE) {
            double dx[NR], dy[NR], dz[NR], far[NR];
            bool hit[NR];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int k = k0 + r * T::SIZE < P ? k0 + r * T::SIZE : k0;
                sp.point(k, &dx[r], &dy[r], &dz[r]);
            }
            if (INL) ray_scan_multi_impl<NR, true>(sh.S, n, cen, dx, dy, dz, hit, far);
            else ray_scan_multi<NR>(sh.S, n, cen, dx, dy, dz, hit, far);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                int k = k0 + r * T::SIZE;
                if (k < P) { vals[k] = far[r]; flag[k] = hit[r] ? 1 : 0; }
            }
        }
    } else {
        for (int k = T::tid(); k < P; k += T::SIZE) {
            double px, py, pz, far;
            sp.point(k, &px, &py, &pz);
            bool hit = ray_scan(sh.S, n, cen, px, py, pz, &far);
            vals[k] = far;
            flag[k] = hit ? 1 : 0;
        }
    }
    T::sync();
    if (T::wave() == 0) PW_T1(ws, 28, t_a1);
    PW_T0(t_a2);
    // compact in ray order (thread 0), then the numpy mean
    // order-preserving compaction by wave 0 (ballot + prefix popcount)
    if (T::wave() == 0) {
        int m = 0;
        for (int base = 0; base < P; base += T::WSIZE) {
            int k = base + T::lane();
            bool f = k < P && flag[k] != 0;
            unsigned long long bal = T::ballot(f);
            int pos = m + __builtin_popcountll(bal & ((1ull << T::lane()) - 1ull));
            if (f) packed[pos] = vals[k];
            m += __builtin_popcountll(bal);
        }
        if (T::lane() == 0) v.n_surv = m;
    }
    T::sync();
    int m = v.n_surv;
    double sum = np_sum_team<T>(packed, m, s_tab, s_acc, s_leaf, &v.red_v[15]);
    if (T::wave() == 0) PW_T1(ws, 29, t_a2);
    if (T::tid() == 0) {
        out->avg_d = (sum / (double)m) * 2.0;
        out->n_points_avg = P;
    }
    T::sync();
}

template <class T>
PW_NOINLINE PW_HD inline void stage_average(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                             const pw_params& prm) {
    PW_ASSUME_TEAM_STATE(sh, prm);
    stage_average_impl<T, false>(sh, ws, n, out, prm);
}

// ---- Nelder-Mead in the window plane (scipy.optimize.fmin defaults, N = 2) ------------------
template <class T>
PW_NOINLINE PW_HD inline void wave_fmin_xy(const Frame& F, int n, double z, double x0, double y0, double* xo,
                               double* yo, int* n_eval) {
    const int maxfun = 400, maxiter = 400;
    const double xatol = 1e-4, fatol = 1e-4;
    // the simplex lives in named scalars: private arrays indexed at run time would be
    // placed in scratch (global) memory on the GPU
    double x0s, y0s, f0s, x1s, y1s, f1s, x2s, y2s, f2s;
    int fcalls = 0;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
    NearGap1<T> near_gap;
    near_gap.init();
    const Frame Fl = F;          // (by value: `F` is behind a generic pointer)
    auto fun = [&](double x, double y) {
        fcalls += 1;
        if (T::WSIZE == 64) return -(near_gap.eval(Fl, n, nullptr, x, y, z) * 2.0);
        return -(wave_gap_value<T>(Fl, n, x, y, z) * 2.0);
    };
#else
    auto fun = [&](double x, double y) {
        fcalls += 1;
        return -(wave_gap_value<T>(F, n, x, y, z) * 2.0);
    };
#endif
    x0s = x0; y0s = y0;
    x1s = (x0 != 0.0) ? (1.0 + 0.05) * x0 : 0.00025; y1s = y0;
    x2s = x0; y2s = (y0 != 0.0) ? (1.0 + 0.05) * y0 : 0.00025;
    f0s = fun(x0s, y0s);
    f1s = fun(x1s, y1s);
    f2s = fun(x2s, y2s);
    // stable insertion sort of three (numpy argsort on 3 elements)
    auto swap01 = [&]() { double t; t = f0s; f0s = f1s; f1s = t; t = x0s; x0s = x1s; x1s = t; t = y0s; y0s = y1s; y1s = t; };
    auto swap12 = [&]() { double t; t = f1s; f1s = f2s; f2s = t; t = x1s; x1s = x2s; x2s = t; t = y1s; y1s = y2s; y2s = t; };
    auto sort3 = [&]() {
        if (f0s > f1s) swap01();
        if (f1s > f2s) { swap12(); if (f0s > f1s) swap01(); }
    };
    sort3();
    int iterations = 1;
    while (fcalls < maxfun && iterations < maxiter) {
        double dx1 = pw_abs(x1s - x0s), dx2 = pw_abs(x2s - x0s);
        double dy1 = pw_abs(y1s - y0s), dy2 = pw_abs(y2s - y0s);
        double mx = pw_max(pw_max(dx1, dy1), pw_max(dx2, dy2));
        double mf = pw_max(pw_abs(f0s - f1s), pw_abs(f0s - f2s));
        if (mx <= xatol && mf <= fatol) break;
        double bx = (x0s + x1s) / 2.0, by = (y0s + y1s) / 2.0;
        double xr = 2.0 * bx - x2s, yr = 2.0 * by - y2s;
        bool over = false;  // _MaxFuncCallError
        auto guarded = [&](double x, double y, double* f) {
            if (fcalls >= maxfun) { over = true; return false; }
            *f = fun(x, y);
            return true;
        };
        double fxr;
        if (guarded(xr, yr, &fxr)) {
            bool doshrink = false;
            if (fxr < f0s) {
                double xe = 3.0 * bx - 2.0 * x2s, ye = 3.0 * by - 2.0 * y2s;
                double fxe;
                if (guarded(xe, ye, &fxe)) {
                    if (fxe < fxr) { x2s = xe; y2s = ye; f2s = fxe; }
                    else { x2s = xr; y2s = yr; f2s = fxr; }
                }
            } else if (fxr < f1s) {
                x2s = xr; y2s = yr; f2s = fxr;
            } else if (fxr < f2s) {
                double xc = 1.5 * bx - 0.5 * x2s, yc = 1.5 * by - 0.5 * y2s;
                double fxc;
                if (guarded(xc, yc, &fxc)) {
                    if (fxc <= fxr) { x2s = xc; y2s = yc; f2s = fxc; }
                    else doshrink = true;
                }
            } else {
                double xcc = 0.5 * bx + 0.5 * x2s, ycc = 0.5 * by + 0.5 * y2s;
                double fxcc;
                if (guarded(xcc, ycc, &fxcc)) {
                    if (fxcc < f2s) { x2s = xcc; y2s = ycc; f2s = fxcc; }
                    else doshrink = true;
                }
            }
            if (doshrink && !over) {
                x1s = x0s + 0.5 * (x1s - x0s);
                y1s = y0s + 0.5 * (y1s - y0s);
                double f;
                if (guarded(x1s, y1s, &f)) f1s = f;
                if (!over) {
                    x2s = x0s + 0.5 * (x2s - x0s);
                    y2s = y0s + 0.5 * (y2s - y0s);
                    if (guarded(x2s, y2s, &f)) f2s = f;
                }
            }
            if (!over) iterations += 1;
        }
        sort3();
    }
    *xo = x0s;
    *yo = y0s;
    *n_eval += fcalls;
}

// The objective of the neck search along z (utilities.py:1296-1303: minimise the diameter at (xo, yo, z)) with its
// forward-difference derivative, as PoreObjective does for the pore centre.
template <class T>
struct NeckObjective {
    const Frame& R;
    int n;
    double xo, yo, lo, up;
    PW_LDS int* cand;
    bool have_last;
    double lz, lf, lg;
    int nfev;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
    NearGap4<T> near_gap;
#endif
    PW_HD NeckObjective(const Frame& R_, int n_, double xo_, double yo_, double lo_, double up_, PW_LDS int* cand_)
        : R(R_), n(n_), xo(xo_), yo(yo_), lo(lo_), up(up_), cand(cand_), have_last(false), lz(0.0), lf(0.0), lg(0.0), nfev(0) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
        near_gap.init();
#endif
    }
    PW_HD __attribute__((always_inline)) void operator()(const double* xq, double& fo, double* go) {
        const double zc = xq[0];
        if (!(have_last && zc == lz)) {
            double h = fd_step(zc, lo, up);
            double z1 = zc + h;
            double dz = z1 - zc;
            double zx[4] = {xo, xo, xo, xo}, zy[4] = {yo, yo, yo, yo}, zz[4] = {zc, z1, zc, z1}, gv[4];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_NEAR_GAP)
            if (T::WSIZE == 64) near_gap.eval(R, n, cand, zx, zy, zz, gv);
            else
#endif
            wave_gap4<T>(R, n, zx, zy, zz, gv);
            double f0 = gv[0] * 2.0;
            double f1 = gv[1] * 2.0;
            nfev += 2;
            lf = f0;
            lg = (f1 - f0) / dz;
            lz = zc;
            have_last = true;
        }
        fo = lf;
        go[0] = lg;
    }
};

// ---- the 20 x 20 grid of a window fit, bounded from above first ------------------------------------------------
// scipy.optimize.brute evaluates the window's objective f = -2 min_i gap_i(p) at 400 points of the plane z = zopt and
// hands the FIRST point with the lowest f to the simplex search (utilities.py:1307-1314): only that point's index is
// used, i.e. the point with the largest clearance m(p) = min_i gap_i(p), lowest index among equals.  The plain form
// evaluates every atom at every point (67 000 pairs per window).  Here:
//   1. the atoms nearest to the hole -- gap at the grid's centre within TAU of the smallest: the ring of a window,
//      a dozen or two -- are copied (stored order, so still grouped by radius) into `scratch`, the window's
//      optimiser block, idle between the neck search and the simplex;
//   2. every point gets an UPPER bound u(p) >= m(p): the minimum over those atoms alone (the same per-atom numbers,
//      so the bound is exact arithmetic, not an estimate) -- a lane's seven points share every read;
//   3. the point with the largest bound is evaluated over ALL atoms: L = m(that point), a LOWER bound of the
//      maximum;
//   4. only points with u(p) >= L can reach the maximum -- the ring atoms ARE what limits the clearance near a
//      window, so that is a point or two; each is evaluated over all atoms, largest value / lowest index wins.
// Nothing here depends on how well the ring was chosen except the number of points left in step 4 (more than 24: the
// caller evaluates the plain form).  Returns false (nothing done) for ungrouped radii, more than 256 atoms per
// lane-pass or a ring that does not fit.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
PW_NOINLINE __device__ inline bool wave_brute_bounded(Frame R, int n, PW_LDS double* scratch, int cap, double zopt,
                                                      double gstart, double gstep, int* gidx_out) {
#ifndef PW_GRID_TAU
#define PW_GRID_TAU 0.7     // (measured on CC3: 0.6 - 1.0 within 10 %, 0.4 and 1.6 slower, 0.2 leaves too many points)
#endif
    constexpr double TAU = PW_GRID_TAU;
    const PW_LDS ClassInfo* C = R.cls;
    const int kk = T::uniform_i(C->k);
    cap = T::uniform_i(cap);
    n = T::uniform_i(n);
    if (kk == 0 || cap < 16 || n > 256) return false;
    PW_LDS double *sx = scratch, *sy = scratch + cap, *sz = scratch + 2 * cap, *sq = scratch + 3 * cap;
    const int lane = T::lane();
    // 1. every atom at the centre of the grid; a lane keeps its (up to four) atoms
    const double cx = 9.5 * gstep + gstart, cy = cx;
    const double ppc = sq3(cx, cy, zopt);
    double ax[4], ay[4], az[4], aq[4], ag[4];
    double b = PW_INF;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        const int ic = i < n ? i : 0;
        ax[t] = R.x[ic]; ay[t] = R.y[ic]; az[t] = R.z[ic]; aq[t] = R.xx[ic];
        const double gg = pw_fma(az[t], zopt, pw_fma(ax[t], cx, ay[t] * cy));
        const double d2 = pw_m2add(gg, aq[t]) + ppc;
        ag[t] = i < n ? pw_sqrt(d2 > 0.0 ? d2 : 0.0) - R.vdw[ic] : PW_INF;
        b = __builtin_fmin(b, ag[t]);
    }
    const double lim = T::wave_min(b) + TAU;
    if (!(lim < PW_INF)) return false;
    int coff[PW_KCLS + 1];
#pragma unroll
    for (int g = 0; g <= PW_KCLS; ++g) coff[g] = 0;
    int total = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (64 * t >= n) continue;
        const bool in = ag[t] <= lim;
        const unsigned long long mk = T::ballot(in);
        const int at = total + (int)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
        if (in && at < cap) { sx[at] = ax[t]; sy[at] = ay[t]; sz[at] = az[t]; sq[at] = aq[t]; }
        // where the groups start in the list: listed atoms stored before a group's first atom
#pragma unroll
        for (int g = 0; g <= PW_KCLS; ++g) {
            const int first = g < kk ? T::uniform_i(C->off[g]) : n;            // (g >= kk: the end of the list)
            const int below = first - 64 * t;                                  // atoms of this pass stored before it
            const unsigned long long lt = below >= 64 ? ~0ull : (below <= 0 ? 0ull : ((1ull << below) - 1ull));
            coff[g] += (int)__builtin_popcountll(mk & lt);
        }
        total += (int)__builtin_popcountll(mk);
    }
    if (total > cap || total == 0) return false;
    T::wave_sync();
    // 2. upper bounds: a lane's seven points over the ring
    double qx[7], qy[7], pp[7], ub[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        int q = lane + 64 * p;
        if (q >= 400) q = lane;
        qx[p] = (double)(q / 20) * gstep + gstart;
        qy[p] = (double)(q % 20) * gstep + gstart;
        pp[p] = sq3(qx[p], qy[p], zopt);
        ub[p] = PW_INF;
    }
#pragma unroll
    for (int g = 0; g < PW_KCLS; ++g) {
        if (g >= kk) continue;
        double m2[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) m2[p] = PW_INF;
        const int jlo = coff[g], jhi = coff[g + 1];
        for (int j = jlo; j < jhi; ++j) {
            const double x = sx[j], y = sy[j], z = sz[j], xx = sq[j];
#pragma unroll
            for (int p = 0; p < 7; ++p) {
                const double gg = pw_fma(z, zopt, pw_fma(x, qx[p], y * qy[p]));
                m2[p] = __builtin_fmin(m2[p], pw_m2add(gg, xx));
            }
        }
        const double r = C->vdw[g];
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            const double m2p = m2[p] + pp[p];
            const double d = pw_sqrt(m2p > 0.0 ? m2p : 0.0);
            ub[p] = __builtin_fmin(ub[p], d - r);
        }
    }
    // 3. the point with the largest bound (lowest index among equals), over all atoms
    double ubest = -PW_INF;
    int uq = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        const int q = lane + 64 * p;
        if (q < 400 && ub[p] > ubest) { ubest = ub[p]; uq = q; }
    }
    {
        double neg = -ubest;                  // (argmin of the negated bound: smallest index among equals)
        T::wave_argmin(neg, uq);
        ubest = -neg;
    }
    if (!(ubest < PW_INF) || uq >= 400) return false;
    double best = wave_gap_value<T>(R, n, (double)(uq / 20) * gstep + gstart, (double)(uq % 20) * gstep + gstart, zopt);
    int bestq = uq;
    // 4. what can still reach it
    int left = 0;
    unsigned long long sv[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        const int q = lane + 64 * p;
        sv[p] = T::ballot(q < 400 && q != uq && ub[p] >= best);
        left += (int)__builtin_popcountll(sv[p]);
    }
    if (left > 24) return false;
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        unsigned long long mk = sv[p];
        while (mk) {
            const int q = (int)__builtin_ctzll(mk) + 64 * p;
            mk &= mk - 1ull;
            const double m = wave_gap_value<T>(R, n, (double)(q / 20) * gstep + gstart, (double)(q % 20) * gstep + gstart, zopt);
            if (m > best || (m == best && q < bestq)) { best = m; bestq = q; }
        }
    }
    *gidx_out = bestq;
    return true;
}
#endif

// ---- the refined path scan of a window fit over the atoms around the path -------------------------------------
// window_analysis walks the cluster's vector in steps of increment2 (0.1 A: ~116 points, utilities.py:1226) and needs
// three things of the walk: that every point has a positive gap, the smallest gap m*, and the FIRST point that has it.
// All three are decided by the atoms that can come within m* of some point of the path.  m* is at most the gap T0 at any
// single point (four points of the outer half are evaluated over all atoms), and an atom's gap at a point of the
// segment is at least its distance to the segment minus its radius: so the atoms with dist(a, segment) - r <= T0 (+ a
// margin a million times the rounding error) contain every atom that attains m* anywhere -- the walk over those alone
// has the same minimum at the same points (elsewhere its values are upper bounds, which is all a minimum needs).
// For a cage that is the rim of one window and the wall around the path, a fifth of the atoms.  `scratch`: the
// window's optimiser block, not yet in use.  Returns false (nothing done) when the radii are ungrouped, the list does
// not fit or a value is not finite: the caller walks the path over all atoms.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
PW_NOINLINE __device__ inline bool wave_path_tube(Frame F, int n, PW_LDS double* scratch, int cap, double cx, double cy,
                                                  double cz, int chunks, double* pbest_out, int* ppos_out, bool* ok_out) {
    const PW_LDS ClassInfo* C = F.cls;
    const int kk = T::uniform_i(C->k);
    cap = T::uniform_i(cap);
    n = T::uniform_i(n);
    chunks = T::uniform_i(chunks);
    if (kk == 0 || cap < 16 || n > 256 || chunks < 16) return false;
    PW_LDS double *sx = scratch, *sy = scratch + cap, *sz = scratch + 2 * cap, *sq = scratch + 3 * cap;
    const int lane = T::lane();
    // 1. an upper bound of the smallest gap: four points of the outer half, every atom
    double T0;
    {
        double px[4], py[4], pz[4], gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = (int)(((long)chunks * (4 + q)) / 8);
            px[q] = cx * (double)k; py[q] = cy * (double)k; pz[q] = cz * (double)k;
        }
        wave_gap4<T>(F, n, px, py, pz, gv);
        T0 = __builtin_fmin(__builtin_fmin(gv[0], gv[1]), __builtin_fmin(gv[2], gv[3]));
    }
    if (!(pw_abs(T0) < 1.0e6)) return false;
    // 2. the atoms within T0 + radius of the segment from the origin to the last point, copied in stored order
    const double ex = cx * (double)chunks, ey = cy * (double)chunks, ez = cz * (double)chunks;
    const double len2 = sq3(ex, ey, ez);
    if (!(len2 > 0.0) || !(len2 < 1.0e12)) return false;
    int coff[PW_KCLS + 1];
#pragma unroll
    for (int g = 0; g <= PW_KCLS; ++g) coff[g] = 0;
    int total = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (64 * t >= n) continue;
        const int i = lane + 64 * t;
        const int ic = i < n ? i : 0;
        const double ax = F.x[ic], ay = F.y[ic], az = F.z[ic], aq = F.xx[ic];
        double s = (ax * ex + ay * ey + az * ez) / len2;
        s = s < 0.0 ? 0.0 : (s > 1.0 ? 1.0 : s);
        const double dx = ax - s * ex, dy = ay - s * ey, dz = az - s * ez;
        const double d2 = dx * dx + dy * dy + dz * dz;
        const double lim = T0 + F.vdw[ic] + 1e-6;
        const bool in = i < n && lim >= 0.0 && d2 <= lim * lim * (1.0 + 1e-9);
        const unsigned long long mk = T::ballot(in);
        const int at = total + (int)__builtin_popcountll(mk & ((1ull << lane) - 1ull));
        if (in && at < cap) { sx[at] = ax; sy[at] = ay; sz[at] = az; sq[at] = aq; }
#pragma unroll
        for (int g = 0; g <= PW_KCLS; ++g) {
            const int first = g < kk ? T::uniform_i(C->off[g]) : n;
            const int below = first - 64 * t;
            const unsigned long long lt = below >= 64 ? ~0ull : (below <= 0 ? 0ull : ((1ull << below) - 1ull));
            coff[g] += (int)__builtin_popcountll(mk & lt);
        }
        total += (int)__builtin_popcountll(mk);
    }
    if (total > cap || total == 0) return false;
    T::wave_sync();
    // 3. the walk over the list: two points per lane and pass, point_gap_value's operations
    double pbest = PW_INF;
    int ppos = 0x7fffffff;
    bool ok = true;
    for (int k0 = lane; k0 <= chunks; k0 += 128) {
        const int k1 = k0 + 64 <= chunks ? k0 + 64 : k0;
        const double qx[2] = {cx * (double)k0, cx * (double)k1};
        const double qy[2] = {cy * (double)k0, cy * (double)k1};
        const double qz[2] = {cz * (double)k0, cz * (double)k1};
        double best[2] = {PW_INF, PW_INF};
        const double pp[2] = {sq3(qx[0], qy[0], qz[0]), sq3(qx[1], qy[1], qz[1])};
#pragma unroll
        for (int g = 0; g < PW_KCLS; ++g) {
            if (g >= kk) continue;
            double m2[2] = {PW_INF, PW_INF};
            const int jlo = coff[g], jhi = coff[g + 1];
            int j = jlo;
            for (; j + 4 <= jhi; j += 4) {
                double bx[4], by[4], bz[4], bq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { bx[t] = sx[j + t]; by[t] = sy[j + t]; bz[t] = sz[j + t]; bq[t] = sq[j + t]; }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const double gg = pw_fma(bz[t], qz[p], pw_fma(bx[t], qx[p], by[t] * qy[p]));
                        m2[p] = __builtin_fmin(m2[p], pw_m2add(gg, bq[t]));
                    }
            }
            for (; j < jhi; ++j) {
                const double bx = sx[j], by = sy[j], bz = sz[j], bq = sq[j];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const double gg = pw_fma(bz, qz[p], pw_fma(bx, qx[p], by * qy[p]));
                    m2[p] = __builtin_fmin(m2[p], pw_m2add(gg, bq));
                }
            }
            const double r = C->vdw[g];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const double m2p = m2[p] + pp[p];
                const double d = pw_sqrt(m2p > 0.0 ? m2p : 0.0);
                best[p] = __builtin_fmin(best[p], d - r);
            }
        }
        if (!(best[0] > 0.0)) ok = false;
        if (best[0] < pbest) { pbest = best[0]; ppos = k0; }
        if (k1 != k0) {
            if (!(best[1] > 0.0)) ok = false;
            if (best[1] < pbest) { pbest = best[1]; ppos = k1; }
        }
    }
    T::wave_sync();
    *pbest_out = pbest;
    *ppos_out = ppos;
    *ok_out = ok;
    return true;
}
#endif

// ---- one window (utilities.py:1191-1361), executed by ONE wave -------------------------------
// per-cluster arrays of the window fits: in the team's LDS (UnitVars) for up to PW_W_MAX clusters, in the
// team's global slab beyond -- the number of clusters has no upper limit (utilities.py:1481-1536)
struct WinArrays {
    double* vec;   // 3 per cluster: the sampling vector with the largest 2*gap
    double* d;     // diameter
    double* c;     // 3 per cluster: centre
    int* ok;       // 1 fitted, 0 dropped by the refined path scan, -1 inverted z bounds
};
// FS: the molecule with the pore centre at the origin; R: where the rotated copy goes -- R may be FS itself (a
// fit worker that owns its frame rotates in place: every lane reads its atoms before it writes them, and FS is
// not read again).  shift: what make_shifted subtracted (added back to the window centre).  evals_sink: lane 0
// adds the number of objective evaluations there.
template <class T>
PW_NOINLINE PW_HD inline void wave_window(const Frame& FS, const Frame& R, PW_LDS void* lbmem, TeamWorkspace* ws, int n,
                                          int cluster, const double* shift, const pw_params& prm, const WinArrays& wa,
                                          int* evals_sink) {
    int evals = 0;
    // (i) the cluster's vector with the largest 2*gap was selected by stage_windows
    double vx = wa.vec[3 * cluster], vy = wa.vec[3 * cluster + 1], vz = wa.vec[3 * cluster + 2];
    PW_T0(t_p);
    // (ii) refined path scan, increment2 (0.1), lanes over path points
    double nrm = norm3(vx, vy, vz);
    int chunks = (int)np_floordiv(nrm, prm.increment2);
    double cx = vx / (double)chunks, cy = vy / (double)chunks, cz = vz / (double)chunks;
    double pbest = PW_INF;
    int ppos = 0x7fffffff;
    bool ok = true;
    bool tubed = false;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_PATH_TUBE)
    if (T::WSIZE == 64)
        tubed = wave_path_tube<T>(FS, n, (PW_LDS double*)lbmem, (int)(sizeof(LbMem<1>) / 8 / 4), cx, cy, cz, chunks, &pbest, &ppos, &ok);
#endif
    if (tubed) {
        // (done: the walk over the atoms around the path, wave_path_tube)
    } else if (T::WSIZE == 64) {
        // two path points per lane and pass over the atoms
        for (int k0 = T::lane(); k0 <= chunks; k0 += 2 * T::WSIZE) {
            int k1 = k0 + T::WSIZE <= chunks ? k0 + T::WSIZE : k0;
            double qx[2] = {cx * (double)k0, cx * (double)k1};
            double qy[2] = {cy * (double)k0, cy * (double)k1};
            double qz[2] = {cz * (double)k0, cz * (double)k1};
            double m[2];
            points_gap_values<2>(FS, n, qx, qy, qz, m);
            if (!(m[0] > 0.0)) ok = false;
            if (m[0] < pbest) { pbest = m[0]; ppos = k0; }
            if (k1 != k0) {
                if (!(m[1] > 0.0)) ok = false;
                if (m[1] < pbest) { pbest = m[1]; ppos = k1; }
            }
        }
    } else {
        for (int k = T::lane(); k <= chunks; k += T::WSIZE) {
            double m = point_gap_value(FS, n, cx * (double)k, cy * (double)k, cz * (double)k);
            if (!(m > 0.0)) ok = false;
            if (m < pbest) { pbest = m; ppos = k; }
        }
    }
    evals += chunks + 1;
    ok = T::wave_all(ok);
    if (!ok) {
        if (T::lane() == 0) {
            wa.ok[cluster] = 0;
            *evals_sink += evals;
        }
        return;
    }
    T::wave_argmin(pbest, ppos);
    PW_T1(ws, 2, t_p);
    PW_T0(t_r);
    double new_z = norm3(cx * (double)ppos, cy * (double)ppos, cz * (double)ppos);
    // (iii) rotation angles (utilities.py:1235-1259)
    // angle_between_vectors (utilities.py:1088-1097): x[i] ** 2 on float64 scalars is libm's pow
    // (A wave computes these scalars in every lane alike.  The library functions among them -- three squares through
    // pow, two arc cosines, two sines and cosines -- go through ONE evaluation each, their arguments side by side in the
    // first lanes: the same instructions on the same values, a third of the dependent chain.)
    double vx2, vy2, vz2;
    if (T::WSIZE == 64) {
        const double sq = pw_square_np(T::lane() == 1 ? vy : (T::lane() == 2 ? vz : vx));
        vx2 = T::bcast_u(sq, 0); vy2 = T::bcast_u(sq, 1); vz2 = T::bcast_u(sq, 2);
    } else {
        vx2 = pw_square_np(vx); vy2 = pw_square_np(vy); vz2 = pw_square_np(vz);
    }
    double c1 = pw_abs(vx * 1.0 + vy * 0.0 + 0.0 * 0.0) /
                (pw_sqrt(vx2 + vy2 + 0.0) * pw_sqrt(1.0 + 0.0 + 0.0));
    double c2 = pw_abs(vx * 0.0 + vy * 0.0 + vz * 1.0) /
                (pw_sqrt(vx2 + vy2 + vz2) * pw_sqrt(0.0 + 0.0 + 1.0));
    double a1, a2;
    if (T::WSIZE == 64) {
        const double ac = pw_acos_np(T::lane() == 1 ? c2 : c1, ws->rsq);
        a1 = T::bcast_u(ac, 0); a2 = T::bcast_u(ac, 1);
    } else {
        a1 = pw_acos_np(c1, ws->rsq); a2 = pw_acos_np(c2, ws->rsq);
    }
    const double a1_raw = a1, a2_raw = a2;       // what angle_between_vectors returned (stage capture)
    bool sxp = vx >= 0.0, syp = vy >= 0.0, szp = vz >= 0.0;
    if (szp) {
        if (sxp && syp) { a1 = -a1; a2 = -a2; }
        else if (!sxp && syp) { a1 = TWO_PI + a1; }
        else if (sxp && !syp) { a2 = -a2; }
        else { a1 = TWO_PI - a1; }
    } else {
        if (sxp && syp) { a1 = -a1; a2 = ONE_PI + a2; }
        else if (!sxp && syp) { a2 = ONE_PI - a2; }
        else if (sxp && !syp) { a2 = a2 + ONE_PI; }
        else { a1 = -a1; a2 = ONE_PI - a2; }
    }
    double s1, co1, s2, co2;
    if (T::WSIZE == 64) {
        double sv, cv;
        pw_sincos(T::lane() == 1 ? a2 : a1, &sv, &cv);
        s1 = T::bcast_u(sv, 0); co1 = T::bcast_u(cv, 0); s2 = T::bcast_u(sv, 1); co2 = T::bcast_u(cv, 1);
    } else {
        pw_sincos(a1, &s1, &co1);
        pw_sincos(a2, &s2, &co2);
    }
    // rows of a 3x3 matrix times a vector through BLAS dgemv: fma(m2,v2, fma(m0,v0, m1*v1))
    auto row = [](double m0, double m1, double m2, double x, double y, double z) {
        return pw_fma(m2, z, pw_fma(m0, x, m1 * y));
    };
    for (int i = T::lane(); i < n; i += T::WSIZE) {
        double x = FS.x[i], y = FS.y[i], z = FS.z[i];
        double x1 = row(co1, -s1, 0.0, x, y, z);
        double y1 = row(s1, co1, 0.0, x, y, z);
        double z1 = row(0.0, 0.0, 1.0, x, y, z);
        double x2 = row(co2, 0.0, s2, x1, y1, z1);
        double y2 = row(0.0, 1.0, 0.0, x1, y1, z1);
        double z2 = row(-s2, 0.0, co2, x1, y1, z1);
        x2 = x2 - 0.0; y2 = y2 - 0.0; z2 = z2 - new_z;
        R.x[i] = x2; R.y[i] = y2; R.z[i] = z2;
        R.xx[i] = sq3(x2, y2, z2);
    }
    T::wave_sync();
    // (iv) diameter at the neck
    double d0 = wave_gap_value<T>(R, n, 0.0, 0.0, 0.0) * 2.0;
    evals += 1;
    PW_T1(ws, 3, t_r);
    // (v) neck position along z: L-BFGS-B, n = 1, bounds [-new_z, +inf) by default
    // (lb_z / z_bounds: utilities.py:1296-1303); (vi) in-plane optimisation; optionally the neck
    // search once more from the in-plane optimum (z_second_mini, :1326-1334)
    Lbfgsb<1> zopt_state;
    Lbfgsb<1>* S = &zopt_state;
    LbMem<1>* Smem = (LbMem<1>*)lbmem;
    PW_ASSUME_LDS(Smem);
    double lo1[1] = {prm.lb_z ? -new_z : prm.z_lo}, up1[1] = {prm.z_hi};
    int nbd1[1];
    {
        bool has_lo = lo1[0] > -PW_INF, has_up = up1[0] < PW_INF;
        nbd1[0] = has_lo ? (has_up ? 2 : 1) : (has_up ? 3 : 0);
    }
    if (lo1[0] > up1[0]) {                       // scipy: ValueError -- reported through the status
        if (T::lane() == 0) {
            wa.ok[cluster] = -1;                 // "bounds", not "path scan failed"
            *evals_sink += evals;
        }
        return;
    }
    double xo = 0.0, yo = 0.0, zopt = 0.0;
    for (int phase = 0; phase < 2; ++phase) {
        if (lo1[0] == up1[0]) {
            zopt = lo1[0];                       // a fixed variable: scipy returns the bound itself
        } else {
            double x01[1] = {zopt};
            // np.clip(x0, lb, ub)
            x01[0] = x01[0] < lo1[0] ? lo1[0] : (x01[0] > up1[0] ? up1[0] : x01[0]);
            S->template setup<T>(Smem, x01, lo1, up1, nbd1, 1e7, 1e-5, 20);
#if defined(PW_PROFILE) && defined(PW_ZPROF)
            S->prof = PW_PROF_BASE(ws);       // (-DPW_ZPROF: the optimiser's timers are the neck search's, not the chains')
#if defined(PW_LB_FINE)
            Smem->prof_fine = PW_PROF_BASE(ws);
#endif
#endif
            int nit = 0;
            // (the candidate list of the objective lives in the optimiser block that setup() just cleared)
            NeckObjective<T> fg(R, n, xo, yo, lo1[0], up1[0], (PW_LDS int*)Smem->cand);
            PW_T0(t_zs);
            S->template minimize<T>(fg, 15000, 15000, &nit);
            T::wave_sync();
            PW_T1(ws, 4, t_zs);
            evals += fg.nfev;
            zopt = S->x[0];
        }
        if (phase == 1) break;
        PW_T0(t_b);
        // (vi) brute 20 x 20 grid over +-d0/2, lanes over grid points, first minimum
        double hlf = d0 / 2.0;
        double gstart = -hlf;
        double gstep = (hlf - gstart) / 19.0;
        double gbest = PW_INF;
        int gidx = 0x7fffffff;
        bool listed = false;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_NO_GRID_LISTS)
        if (T::WSIZE == 64)
            listed = wave_brute_bounded<T>(R, n, (PW_LDS double*)lbmem, (int)(sizeof(LbMem<1>) / 8 / 4), zopt, gstart, gstep, &gidx);
#endif
        if (listed) {
            // (done, and gidx is the same in every lane: the grid bounded from above first, wave_brute_bounded)
            gbest = 0.0;
        } else if (T::WSIZE == 64) {
            // a lane's (up to) seven grid points share one pass over the atoms (PW_TILE_GRID of them at a time)
            constexpr int NP = PW_TILE_GRID;
            for (int p0 = 0; p0 < 7; p0 += NP) {
                double qx[NP], qy[NP], qz[NP], m[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    int q = T::lane() + 64 * (p0 + p);
                    if (q >= 400) q = T::lane();
                    qx[p] = (double)(q / 20) * gstep + gstart;
                    qy[p] = (double)(q % 20) * gstep + gstart;
                    qz[p] = zopt;
                }
                points_gap_values<NP>(R, n, qx, qy, qz, m);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    int q = T::lane() + 64 * (p0 + p);
                    double f = -(m[p] * 2.0);
                    if (p0 + p < 7 && q < 400 && f < gbest) { gbest = f; gidx = q; }
                }
            }
        } else {
            for (int q = T::lane(); q < 400; q += T::WSIZE) {
                int ix = q / 20, iy = q % 20;
                double gx = (double)ix * gstep + gstart, gy = (double)iy * gstep + gstart;
                double f = -(point_gap_value(R, n, gx, gy, zopt) * 2.0);
                if (f < gbest) { gbest = f; gidx = q; }
            }
        }
        evals += 400;
        T::wave_argmin(gbest, gidx);
        double gx0 = (double)(gidx / 20) * gstep + gstart, gy0 = (double)(gidx % 20) * gstep + gstart;
        PW_T1(ws, 6, t_b);
        PW_T0(t_n);
        wave_fmin_xy<T>(R, n, zopt, gx0, gy0, &xo, &yo, &evals);
        PW_T1(ws, 7, t_n);
        if (!prm.z_second_mini) break;
    }
    // (vii) final diameter, (viii) back-rotation
    double dfin = wave_gap_value<T>(R, n, xo, yo, zopt) * 2.0;
    evals += 1;
    double wx = xo, wy = yo, wz = zopt + new_z;
    // (the sine and cosine of -a2 and -a1: the library's sine is odd and its cosine even bit for bit -- every branch of
    // s_sin.c works on |x| and restores the sign, the reduction of a large argument is odd operation by operation;
    // tests/test_math.py holds the restatement to the library on both signs)
    const double sm2 = -s2, cm2 = co2, sm1 = -s1, cm1 = co1;
    double tx = row(cm2, 0.0, sm2, wx, wy, wz);
    double ty = row(0.0, 1.0, 0.0, wx, wy, wz);
    double tz = row(-sm2, 0.0, cm2, wx, wy, wz);
    double ux = row(cm1, -sm1, 0.0, tx, ty, tz);
    double uy = row(sm1, cm1, 0.0, tx, ty, tz);
    double uz = row(0.0, 0.0, 1.0, tx, ty, tz);
#ifdef PW_HOST_DEBUG
    printf("DBG cluster %d vec %.17g %.17g %.17g a1 %.17g a2 %.17g new_z %.17g d0 %.17g zopt %.17g xy %.17g %.17g dfin %.17g\n", cluster, vx, vy, vz, a1, a2, new_z, d0, zopt, xo, yo, dfin);
#endif
    if (T::lane() == 0) {
        if (pw_unit_debug* dbg = (ws->dbg_base && cluster < PW_W_MAX) ? ws->dbg_base + ws->unit : nullptr) {
            double* wd = dbg->win[cluster];
            wd[0] = vx; wd[1] = vy; wd[2] = vz; wd[3] = a1_raw; wd[4] = a2_raw; wd[5] = new_z; wd[6] = d0;
            wd[7] = zopt; wd[8] = xo; wd[9] = yo; wd[10] = dfin; wd[11] = (double)evals;
        }
        wa.ok[cluster] = 1;
        wa.d[cluster] = dfin;
        wa.c[3 * cluster] = ux + shift[0];
        wa.c[3 * cluster + 1] = uy + shift[1];
        wa.c[3 * cluster + 2] = uz + shift[2];
        *evals_sink += evals;
    }
}

// ---- DBSCAN(eps, min_samples = 5) (sklearn.cluster.DBSCAN as find_windows calls it, utilities.py:1478-1487)
// on points pts[PT(surv_k[i], c)], i < ns: labels[i] = sklearn's label.  sklearn (_dbscan_inner.pyx): core
// points = connected components of the eps-graph numbered by their smallest member index; a border point
// takes the label of the first (lowest-numbered) cluster with a core point next to it.  Both are
// independent of the traversal order, so all clusters are found at once.  core / flags / roots: three
// team-shared bit sets of at least ns bits; adjacency rows in the arena (LDS) when they fit, else in the
// team's global workspace.  Returns the number of clusters (every thread), -1 if the rows fit nowhere.
template <class T, class PTF>
PW_HD inline __attribute__((always_inline)) int team_dbscan(PW_LDS unsigned long long* core, PW_LDS unsigned long long* flags,
                                                            PW_LDS unsigned long long* roots, ScratchArena arena,
                                                            TeamWorkspace* ws, const double* pts, PTF PT, const int* surv_k,
                                                            int ns, double eps, int* labels) {
    PW_T0(t_adj);
    const int words = (ns + 63) / 64;
    const double e2 = eps * eps;
    // adjacency rows live in LDS (the window frames are idle now) when they fit
    unsigned long long* adj = (unsigned long long*)arena.take((size_t)ns * (size_t)words * 8);
    int stride = words;
    if (!adj) { adj = ws->adj; stride = ws->p_cap / 64; }
    if (adj == nullptr) return -1;   // launch without a global adjacency buffer and LDS too small
    for (int wd = T::tid(); wd < words; wd += T::SIZE) { core[wd] = 0; roots[wd] = 0; }
    PW_LDS int* chg = (PW_LDS int*)flags;          // "something changed" flags of the rounds below, used in turn
    if (T::tid() == 0) { chg[0] = 0; chg[1] = 0; }
    // the survivors' end points, compacted: the threads of a wave then read the same point at the
    // same time (broadcast) and eight of them are fetched ahead of the arithmetic
    double* cp = (double*)arena.take((size_t)ns * 24);
    for (int i = T::tid(); i < ns; i += T::SIZE) {
        labels[i] = 0;                  // (counts the neighbours first)
        if (cp) {
            const int pik = surv_k[i];
            cp[3 * i] = pts[PT(pik, 0)]; cp[3 * i + 1] = pts[PT(pik, 1)]; cp[3 * i + 2] = pts[PT(pik, 2)];
        }
    }
    T::sync();
    // Work items are (point, 64-bit word of its adjacency row), dealt point-major: the number of
    // survivors is rarely a multiple of the team size (CC3: 250-270 against 256 threads), and a
    // second round of whole rows for a handful of points would cost as much as the first.
#define PW_ROW_ITEMS_BEGIN                                                                   \
    {                                                                                    \
        int i = T::tid(), wd = 0;                                                        \
        while (i >= ns && wd < words) { i -= ns; ++wd; }                                 \
        while (wd < words) {
#define PW_ROW_ITEMS_END                                                                     \
            i += T::SIZE;                                                                \
            while (i >= ns && wd < words) { i -= ns; ++wd; }                             \
        }                                                                                \
    }
    const bool have_cp = cp != nullptr;
    // (are the points in order of falling z?  The survivors of the window search are -- ray order; a caller's cloud need
    // not be: pw_dbscan.  chg[1] is free until the second round of the propagation below, which clears it first.)
    if (have_cp) {
        for (int i = T::tid(); i + 1 < ns; i += T::SIZE)
            if (!(cp[3 * i + 2] >= cp[3 * (i + 1) + 2])) chg[1] = 1;
    }
    T::sync();
#ifndef PW_NO_SCAN_LISTS
    const bool z_falls = have_cp && chg[1] == 0;
#else
    const bool z_falls = false;
#endif
    auto cluster = [&](auto cp_, auto adj_, auto labels_) __attribute__((always_inline)) {
        // (the rows are computed HALF a word -- 32 candidate points -- to a work item: 540 whole-word items on 256 threads are
        // three rounds for one wave and two for the others, 1080 halves are five against four: 2.5 words, not 3)
        {
            using A32 = std::conditional_t<std::is_same<decltype(adj_), unsigned long long*>::value, unsigned*, PW_LDS unsigned*>;
            A32 adj32 = (A32)adj_;
            const int nh = 2 * words;
            int i = T::tid(), h = 0;
            while (i >= ns && h < nh) { i -= ns; ++h; }
            while (h < nh) {
                double px, py, pz;
                if (have_cp) { px = cp_[3 * i]; py = cp_[3 * i + 1]; pz = cp_[3 * i + 2]; }
                else { const int pik = surv_k[i]; px = pts[PT(pik, 0)]; py = pts[PT(pik, 1)]; pz = pts[PT(pik, 2)]; }
                unsigned bits = 0;
                const int j0 = h * 32;
                const int jend = j0 + 32 < ns ? j0 + 32 : ns;
                int j = j0;
                // Points in order of falling z (the survivors: ray order): the z of the 32 candidates lie between those of
                // the first and the last.  A block whose z are all farther than eps from this point's
                // holds no neighbour (d >= dz * dz > eps^2, a part in 1e9 to spare): two reads instead of 32 tests,
                // for two blocks in three (16 -> 6 us of the stage; the bits are the same).
                if (z_falls && j0 < jend) {
                    const double za = cp_[3 * j0 + 2], zb = cp_[3 * (jend - 1) + 2];
                    const double zhi = za > zb ? za : zb, zlo = za > zb ? zb : za;
                    const double reach = eps * (1.0 + 1e-9);
                    if (pz - zhi > reach || zlo - pz > reach) j = jend;
                }
                if (have_cp) {
                    for (; j + 8 <= jend; j += 8) {
                        double qx[8], qy[8], qz[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) { qx[t] = cp_[3 * (j + t)]; qy[t] = cp_[3 * (j + t) + 1]; qz[t] = cp_[3 * (j + t) + 2]; }
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            double ax = px - qx[t], ay = py - qy[t], az = pz - qz[t];
                            double d = ax * ax;      // (0.0 + ax * ax is ax * ax exactly)
                            d = d + ay * ay; d = d + az * az;
                            if (d <= e2) bits |= 1u << (j + t - j0);
                        }
                    }
                }
                for (; j < jend; ++j) {
                    const int pjk = surv_k[j];
                    double ax = px - pts[PT(pjk, 0)], ay = py - pts[PT(pjk, 1)], az = pz - pts[PT(pjk, 2)];
                    double d = ax * ax;
                    d = d + ay * ay; d = d + az * az;
                    if (d <= e2) bits |= 1u << (j - j0);
                }
                adj32[((size_t)i * stride + (h >> 1)) * 2 + (h & 1)] = bits;
                if (bits) team_atomic_add(&labels_[i], __builtin_popcount(bits));
                i += T::SIZE;
                while (i >= ns && h < nh) { i -= ns; ++h; }
            }
        }
        T::sync();
        // core points (at least min_samples = 5 neighbours, itself included) start as their own
        // component; the others carry "none"
        constexpr int NONE = 0x7fffffff;
        for (int i = T::tid(); i < ns; i += T::SIZE) {
            const bool is_core = labels_[i] >= 5;
            if (is_core) team_atomic_or(&core[i >> 6], 1ull << (i & 63));
            labels_[i] = is_core ? i : NONE;
        }
        T::sync();
        if (T::wave() == 0) PW_T1(ws, 11, t_adj);     // adjacency rows + core points
        PW_T0(t_bfs);
        // Components of the core graph by minimum-label propagation, all clusters at once: a core
        // point takes the smallest label among its core neighbours (one row word per work item, the
        // minimum entered atomically), then jumps to the label of the point its label names.  Labels
        // only decrease and never below the smallest index of the component, so a round in which
        // nothing changed was a round over a constant state: every component then carries its
        // smallest member index.  The fixed point does not depend on the order of the updates.
        for (int round = 0;; ++round) {
            PW_LDS int* changed = &chg[round & 1];
            PW_ROW_ITEMS_BEGIN
                unsigned long long b = adj_[(size_t)i * stride + wd] & core[wd];
                if (b != 0 && ((core[i >> 6] >> (i & 63)) & 1ull)) {
                    int m = NONE;
                    // (four neighbours' labels requested together: taken one at a time, every bit of the word was a
                    // round trip to team memory behind the one before)
                    while (b) {
                        const int j0 = wd * 64 + __builtin_ctzll(b);
                        b &= b - 1;
                        int j1 = j0, j2 = j0, j3 = j0;
                        if (b) { j1 = wd * 64 + __builtin_ctzll(b); b &= b - 1; }
                        if (b) { j2 = wd * 64 + __builtin_ctzll(b); b &= b - 1; }
                        if (b) { j3 = wd * 64 + __builtin_ctzll(b); b &= b - 1; }
                        const int l0 = labels_[j0], l1 = labels_[j1], l2 = labels_[j2], l3 = labels_[j3];
                        const int la = l0 < l1 ? l0 : l1, lb = l2 < l3 ? l2 : l3;
                        const int lm = la < lb ? la : lb;
                        m = lm < m ? lm : m;
                    }
                    if (m < labels_[i]) { team_atomic_min(&labels_[i], m); *changed = 1; }
                }
            PW_ROW_ITEMS_END
            T::sync();
            const bool again = *changed != 0;
            if (!again) break;
            if (T::tid() == 0) chg[(round + 1) & 1] = 0;
            for (int i = T::tid(); i < ns; i += T::SIZE) {
                const int l = labels_[i];
                if (l != NONE) { const int r = labels_[l]; if (r < l) labels_[i] = r; }
            }
            T::sync();
        }
        // clusters are numbered by their smallest member (sklearn visits the points in index order)
        for (int i = T::tid(); i < ns; i += T::SIZE)
            if (labels_[i] == i) team_atomic_or(&roots[i >> 6], 1ull << (i & 63));
        // a border point belongs to the lowest-numbered cluster with a core point next to it
        // (only the labels of core points are read here, only those of the others written)
        for (int i = T::tid(); i < ns; i += T::SIZE) {
            if ((core[i >> 6] >> (i & 63)) & 1ull) continue;
            int m = NONE;
            for (int wd = 0; wd < words; ++wd) {
                unsigned long long b = adj_[(size_t)i * stride + wd] & core[wd];
                while (b) {
                    const int j = wd * 64 + __builtin_ctzll(b);
                    b &= b - 1;
                    const int lj = labels_[j];
                    m = lj < m ? lj : m;
                }
            }
            labels_[i] = m;
        }
        T::sync();
        for (int i = T::tid(); i < ns; i += T::SIZE) {
            const int r = labels_[i];
            int lab = -1;
            if (r != NONE) {
                lab = __builtin_popcountll(roots[r >> 6] & ((1ull << (r & 63)) - 1ull));
                for (int wd = 0; wd < (r >> 6); ++wd) lab += __builtin_popcountll(roots[wd]);
            }
            labels_[i] = lab;
        }
        T::sync();
        if (T::wave() == 0) PW_T1(ws, 23, t_bfs);     // clusters
    };
    if (have_cp && PW_IS_LDS(cp) && PW_IS_LDS(adj) && PW_IS_LDS(labels)) cluster(PW_AS_LDS(cp), PW_AS_LDS(adj), PW_AS_LDS(labels));
    else cluster(cp, adj, labels);
#undef PW_ROW_ITEMS_BEGIN
#undef PW_ROW_ITEMS_END
    int label = 0;
    for (int wd = 0; wd < words; ++wd) label += __builtin_popcountll(roots[wd]);
    return label;
}

// ---- stage: windows ----------------------------------------------------------------------------
// ---- k-NN of the sampling sphere: the search that needs no table --------------------------------------------
// The ten smallest squared distances (ascending) of the four consecutive sampling vectors 4 grp .. 4 grp + 3 to
// all P vectors: candidates within an index window either side (a spiral lattice has its near neighbours at the
// same index offsets everywhere), four points sharing every candidate read, sorted insertion by a
// compare-exchange chain; a window that cannot prove itself (tenth distance within (W - 1) z-levels) is redone
// over the whole sphere.  Writes the DISTANCES (square roots) to rows[10 k + q].  Out of line: its forty-plus
// forty live doubles are the register peak of the sampling stages, and with the neighbour tables
// (nb_build_point) it runs for no point at all -- only when a table is missing or cannot be proven.
// pts_: the sampling vectors, point k at [(c * 4 + (k & 3)) * Q4 + (k >> 2)], c = 0, 1, 2.
template <class PTS>
PW_NOINLINE PW_HD inline void knn_window_group(PTS pts_, int Q4, int P, int grp, int W, double radius, double zstep,
                                               double* rows) {
    constexpr int NK = 4;
    const int k0 = grp * NK;
    double px[NK], py[NK], pz[NK], t[NK][10];
#pragma unroll
    for (int p = 0; p < NK; ++p) {
        const int k = k0 + p < P ? k0 + p : P - 1;
        const int at = (k & 3) * Q4 + (k >> 2);
        px[p] = pts_[at]; py[p] = pts_[4 * Q4 + at]; pz[p] = pts_[8 * Q4 + at];
    }
    const int klast = k0 + NK - 1 < P ? k0 + NK - 1 : P - 1;
    int lo = k0 - W < 0 ? 0 : k0 - W, hi = klast + W > P - 1 ? P - 1 : klast + W;
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int p = 0; p < NK; ++p)
#pragma unroll
            for (int q = 0; q < 10; ++q) t[p][q] = PW_INF;
        // pass 0: every thread walks the same offsets tt relative to its first point (k0 = 4 * grp), so the lanes
        // of a wave read consecutive doubles of one residue class; pass 1 (the whole sphere, only if the window
        // did not prove itself) covers every index
        const int t_lo = pass == 0 ? -W : -k0, t_hi = pass == 0 ? NK - 1 + W : P - 1 - k0;
        // Candidates further than twice the expected tenth-neighbour distance are never entered (pass 0): for
        // most offsets no lane of the wave inserts anything and the whole insertion chain is skipped.  A list
        // that is not full afterwards (t[9] still infinite) fails the proof below and the point is redone
        // without a threshold.
        const double tau = pass == 0 ? 80.0 * radius * radius / (double)P : PW_INF;
        double thr[NK];
#pragma unroll
        for (int p = 0; p < NK; ++p) thr[p] = tau;
PW_PRAGMA(unroll PW_UNROLL_KNN)
        for (int tt = t_lo; tt <= t_hi; ++tt) {
            const int j = k0 + tt;
            if (j < lo || j > hi) continue;
            const int at = ((tt & 3) * Q4) + grp + (tt >> 2);      // j = 4 * grp + tt
            const double qx = pts_[at], qy = pts_[4 * Q4 + at], qz = pts_[8 * Q4 + at];
#pragma unroll
            for (int p = 0; p < NK; ++p) {
                double ax = px[p] - qx, ay = py[p] - qy, az = pz[p] - qz;
                double d = ax * ax;          // (0.0 + ax * ax of the reference is ax * ax exactly)
                d = d + ay * ay; d = d + az * az;
                if (d < thr[p]) {
                    double v_ = d;
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        double lo_ = __builtin_fmin(t[p][q], v_);
                        v_ = __builtin_fmax(t[p][q], v_);
                        t[p][q] = lo_;
                    }
                    thr[p] = __builtin_fmin(tau, t[p][9]);
                }
            }
        }
        bool full = (lo == 0 && hi == P - 1);
        bool proven = true;
#pragma unroll
        for (int p = 0; p < NK; ++p) proven = proven && (pw_sqrt(t[p][9]) < (double)(W - 1) * zstep);
        if (full || proven) break;
        lo = 0; hi = P - 1;
    }
#pragma unroll
    for (int p = 0; p < NK; ++p)
        if (k0 + p < P) {
#pragma unroll
            for (int q = 0; q < 10; ++q) rows[(size_t)(k0 + p) * 10 + q] = pw_sqrt(t[p][q]);
        }
}

// ---- find_windows in two parts (utilities.py:1364-1553) -------------------------------------------------
// windows_bulk: everything up to and including the clustering -- shift, sampling sphere, DBSCAN radius, ray
// pre-analysis, path scans, DBSCAN, the vector chosen for every cluster (:1374-1487, :1221) -- the part that
// is bulk loops over sampling vectors and atoms and needs no optimiser state.  Returns the number of clusters
// (their chosen vectors in wa.vec, wa.ok cleared), or -1 when the search has ended (no vector reaches the
// outside, or a capacity flag: the record is final).  wave_window fits one cluster; windows_finish assembles
// the record.  One team does all three in a row (stage_windows).  (Round 4 also handed the clusters from a sampling
// launch to one-wave fit workers through a ticket per unit: measured 4-25x slower, removed in round 5 -- DESIGN.md
// section 3, profiles/r04_split_*.)
template <class T>
PW_HD inline __attribute__((always_inline)) int windows_bulk_impl(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                                                  const pw_params& prm, WinArrays& wa) {
    PW_ASSUME_TEAM_STATE(sh, prm);
    PW_DCHECK(__builtin_amdgcn_read_exec() == ~0ull, 111);
    auto& v = *sh.v;
    // shift so that the optimised pore centre (pore_opt) or the centre of mass is the origin
    // (utilities.py:1380-1393)
    if (T::tid() == 0) {
        for (int c = 0; c < 3; ++c) {
            double adjust = prm.pore_opt ? v.com[c] - v.opt_c[c] : 0.0;
            v.shift[c] = v.com[c] - adjust;
        }
        for (int w = 0; w < 8; ++w) v.red_i[8 + w] = 0;
    }
    T::sync();
    PW_T0(t_pre);
    make_shifted<T>(sh, n, v.shift[0], v.shift[1], v.shift[2]);
    double keep_d = v.maxd;
    int keep_i = v.maxd_i, keep_j = v.maxd_j;
    T::sync();
    if (T::wave() == 0) PW_T1(ws, 14, t_pre);
    PW_T0(t_md);
    team_max_dim<T, true>(sh, sh.S, n, 2 * n + 2 <= ws->p_cap ? ws->vals : nullptr);
    PW_DCHECK(__builtin_amdgcn_read_exec() == ~0ull, 112);
    double radius = v.maxd / 2.0;
    T::sync();
    if (T::wave() == 0) PW_T1(ws, 15, t_md);
    if (T::tid() == 0) { v.maxd = keep_d; v.maxd_i = keep_i; v.maxd_j = keep_j; }
    if (!(radius / prm.increment < PW_PATH_POINTS_MAX) || !(radius / prm.increment2 < PW_PATH_POINTS_MAX)) {
        // The path of a sampling vector has radius / increment points, a cluster's refined one radius / increment2.  A pore
        // centre that an open or enormous search box let run away (the objective has no maximum out there; the sphere is
        // then as large as its distance) makes the reference build lists of billions of points -- a MemoryError, or hours;
        // here it would be a launch that never ends.  No windows for this unit, and a status that says why.
        if (T::tid() == 0) {
            v.status |= PW_ST_PATH_TOO_LONG;
            out->n_points = 0; out->sphere_r = radius; out->n_windows = -1; out->n_clusters = 0; out->n_survivors = 0; out->eps = 0.0;
        }
        T::sync();
        return -1;
    }
    int P = sampling_count(radius, prm.adjust_windows);
    if (T::tid() == 0) {
        out->n_points = P;
        out->sphere_r = radius;
        out->n_windows = -1;
        out->n_clusters = 0;
        out->n_survivors = 0;
        out->eps = 0.0;
    }
    if (P > ws->p_cap || P < 10) {
        // too many for this launch's workspace (see stage_average), or fewer than the ten neighbours the
        // DBSCAN radius is taken from (the reference: KDTree.query(k=10) raises, utilities.py:1428-1431)
        if (T::tid() == 0) v.status |= P < 10 ? PW_ST_TOO_FEW_POINTS : PW_ST_POINTS_OVERFLOW;
        T::sync();
        return -1;
    }
    PW_DCHECK(radius > 1.0 && radius < 1.0e3, 101);
    Sphere sp;
    sp.init(radius, P);
    // per-unit arrays of the sampling stages: LDS scratch first, global workspace otherwise
    ScratchArena arena;
    arena.init(sh);
    // The sampling vectors, one array per component, each cut into the four residue classes of the
    // point index: point k sits at [component][k & 3][k >> 2].  Threads of a wave that hold
    // consecutive groups of four points and walk their index windows in step (the k-NN loop) then
    // read consecutive doubles -- no LDS bank conflicts -- and everybody else just uses PT().
    const int Q4 = (P + 3) >> 2;
    double* pts = (double*)arena.take((size_t)Q4 * 12 * 8);
    if (!pts) pts = ws->pts;
    auto PT = [Q4](int k, int c) { return (c * 4 + (k & 3)) * Q4 + (k >> 2); };
    // (the per-vector arrays of the later stages are taken after the DBSCAN radius is known: until then
    // everything behind the sampling points belongs to the k-NN mean)
    ScratchArena arena_pts = arena;
    team_sphere_points<T>(ws, sp, [&](int k, double x, double y, double z) { pts[PT(k, 0)] = x; pts[PT(k, 1)] = y; pts[PT(k, 2)] = z; });
    T::sync();
    if (T::wave() == 0) PW_T1(ws, 26, t_pre);
    // ---- eps: mean of all 10-NN distances (self included), utilities.py:1427-1434 ----
    PW_T0(t_eps);
    {
        PW_T0(t_knn);
        // A thread takes FOUR consecutive points (a "group").  With this P's neighbour table (the sixteen nearest
        // points of every point on the unit sphere, NbTables) a point is sixteen exact distances that must come
        // out ascending, the tenth provably below everything outside the list; a point the table cannot prove --
        // or a context without tables -- sends its group through the windowed search (knn_window_group), which
        // leaves the group's distances in the team's global rows.
        constexpr int NK = 4;
        const int ngroups = (P + NK - 1) / NK;
        // scratch of the mean: leaf tables, accumulators, and -- when every thread has at most one group of
        // points -- a tile for the streamed sum, so that the P x 10 distances never leave the CU (they were
        // 64 KB per unit written to and read back from the global workspace)
        int* s_tab = (int*)arena.take(324 * 4);
        double* s_leaf = (double*)arena.take(256 * 8);
        double* s_acc = nullptr;
        double* tile = nullptr;
        int tile_cap = 0;
        if (s_tab && s_leaf && T::SIZE > 1 && ngroups <= T::SIZE) {
            // as large a tile as the arena allows (in 64-element slots, at most 4096 elements), with
            // eight accumulators per slot beside it
            long words = (long)(arena.left / 8) - 160;
            int slots = (int)(words / (64 + 8)) - 2;
            if (slots > 64) slots = 64;
            if (slots >= 4) {
                size_t acc_words = 8 * (size_t)(slots + 2);
                if (acc_words < 128) acc_words = 128;
                s_acc = (double*)arena.take(acc_words * 8);
                tile = (double*)arena.take((size_t)slots * 64 * 8);
                tile_cap = slots * 64 - 128;          // a tile ends at a leaf start at or before lo + tile_cap + 127
            }
        }
        const bool streamed = tile != nullptr && s_acc != nullptr && tile_cap >= 128;
        if (!streamed) {
            arena = arena_pts;
            s_tab = (int*)arena.take(324 * 4);
            s_acc = (double*)arena.take(8 * 160 * 8);
            s_leaf = (double*)arena.take(256 * 8);
            if (!s_tab || !s_acc || !s_leaf) { s_tab = ws->leaf_tab; s_acc = ws->acc8; s_leaf = ws->leaf; }
        }
        // this P's neighbour table, if the context has one
        const unsigned short* nb_idx = nullptr;
        const double* nb_bound = nullptr;
        if (ws->nb_off && P >= PW_NB_PMIN && P <= PW_NB_PMAX && ws->nb_off[P] != PW_NB_NONE) {
            nb_idx = ws->nb_idx + (size_t)ws->nb_off[P] * PW_NB_K;
            nb_bound = ws->nb_bound + ws->nb_off[P];
        }
        // (the windowed search: the 10th neighbour of a point on the equator is about R*sqrt(40/P) away, i.e.
        // 0.112*P z-levels: a window of 0.118*P + 6 indices either side proves itself for every point)
        const int W = (int)(0.118 * (double)P) + 6;
        const double zstep = pw_abs(sp.step) * radius;
        // the ten distances of point k (ascending) from the table; false: the table cannot vouch for them
        auto point_tabled = [&](auto pts_, int k, double* d10) __attribute__((always_inline)) {
            if (!nb_idx) return false;
            const double slack = (radius * radius) * (1.0 - 1e-9);
            const double px = pts_[PT(k, 0)], py = pts_[PT(k, 1)], pz = pts_[PT(k, 2)];
            const unsigned* row = (const unsigned*)(nb_idx + (size_t)k * PW_NB_K);   // (32-byte rows)
            unsigned w[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) w[c] = row[c];
            double prev = -1.0, cd[PW_NB_K];
            bool sorted = true;
#pragma unroll
            for (int c = 0; c < PW_NB_K; ++c) {
                const int j = (int)((w[c >> 1] >> (16 * (c & 1))) & 0xffffu);
                const double qx = pts_[PT(j, 0)], qy = pts_[PT(j, 1)], qz = pts_[PT(j, 2)];
                double ax = px - qx, ay = py - qy, az = pz - qz;
                double d = ax * ax;
                d = d + ay * ay; d = d + az * az;
                sorted = sorted && d >= prev;
                prev = d;
                cd[c] = d;
                if (c < 10) d10[c] = d;
            }
            if (!sorted) {
                // mirror-image neighbours near the equator are ties on the unit sphere that the scaled
                // arithmetic breaks either way: sort what the list holds (rare)
#pragma unroll
                for (int q = 0; q < 10; ++q) d10[q] = PW_INF;
#pragma unroll
                for (int c = 0; c < PW_NB_K; ++c) {
                    double v_ = cd[c];
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        double lo_ = __builtin_fmin(d10[q], v_);
                        v_ = __builtin_fmax(d10[q], v_);
                        d10[q] = lo_;
                    }
                }
            }
            const bool ok = d10[9] < nb_bound[k] * slack;
#pragma unroll
            for (int q = 0; q < 10; ++q) d10[q] = pw_sqrt(d10[q]);
            return ok;
        };
        // the ten distances of point k of group grp into d10; `searched` (per group): the windowed search has
        // run for this group and its rows in ws->knn hold the distances
        auto point_dists = [&](auto pts_, int grp, int k, bool& searched, double* d10) __attribute__((always_inline)) {
            if (!searched && point_tabled(pts_, k, d10)) return;
            if (!searched) {
                knn_window_group(pts_, Q4, P, grp, W, radius, zstep, ws->knn);
                searched = true;
            }
#pragma unroll
            for (int q = 0; q < 10; ++q) d10[q] = ws->knn[(size_t)k * 10 + q];
        };
        if (!streamed) {
            // the distances go to the team's global rows, the mean is taken from there
            auto rows_of_groups = [&](auto pts_) __attribute__((always_inline)) {
                for (int grp = T::tid(); grp < ngroups; grp += T::SIZE) {
                    bool searched = false;
                    for (int p = 0; p < NK; ++p) {
                        const int k = grp * NK + p;
                        if (k >= P) break;
                        double d10[10];
                        point_dists(pts_, grp, k, searched, d10);
                        if (!searched) {
#pragma unroll
                            for (int q = 0; q < 10; ++q) ws->knn[(size_t)k * 10 + q] = d10[q];
                        }
                    }
                }
            };
            if (PW_IS_LDS(pts)) rows_of_groups(PW_AS_LDS(pts)); else rows_of_groups(pts);
        }
        T::sync();
        if (T::wave() == 0) PW_T1(ws, 24, t_knn);
        PW_T0(t_sum);
        double sum;
        if (streamed) {
            // numpy's pairwise sum over the flattened (P, 10) array without the array: the distances go
            // through a tile of team-shared memory, a run of whole leaves of the recursion at a time (a
            // leaf -- at most 128 consecutive elements -- is all a leaf sum needs); the leaf sums are
            // kept and combined in recursion order at the end, chunk by chunk of 8192 elements.
            // A thread's forty distances are formed when the tile that holds them is filled (a group that
            // straddles two tiles is formed twice: sixty-four distance evaluations) -- nothing of the k-NN stage
            // stays in registers across the barriers of the sum.
            const int n_el = P * 10;
            const int grp = T::tid();                       // (streamed: at most one group per thread)
            const int e_first = grp * NK * 10;              // flattened position of this thread's first distance
            bool searched = false;
            double total = 0.0;
            bool first = true;
            for (int s0 = 0; s0 < n_el; s0 += 8192) {
                const int len = n_el - s0 < 8192 ? n_el - s0 : 8192;
                np_leaf_table<T>(len, s_tab);           // (visible after the barrier behind the first tile's fill)
                int lo = 0;
                while (lo < len) {
                    // the tile ends at the start of the leaf that holds element lo + tile_cap (leaf starts
                    // are at most 128 apart, so the tile makes progress), or at the end of the chunk
                    int hi = len;
                    if (lo + tile_cap < len) {
                        int off, l;
                        np_descend(len, lo + tile_cap, &off, &l);
                        hi = off;
                    }
                    auto fill = [&](auto tile_, auto pts_) __attribute__((always_inline)) {
                        if (nb_idx) {
                            // With the neighbour table a point stands alone: the points whose distances fall into this
                            // tile go ONE to a thread (a tile holds a few dozen points; handed out by groups -- four
                            // consecutive points to one thread -- a tile kept a dozen threads busy four points long
                            // and the stage was tiles x that).  A point its table cannot vouch for sends its group
                            // through the windowed search and reads its row back (never seen for CC3).
                            const int E0 = s0 + lo, E1 = s0 + hi;
                            const int k1 = (E1 - 1) / 10;
                            for (int k = E0 / 10 + T::tid(); k <= k1; k += T::SIZE) {
                                double d10[10];
                                if (!point_tabled(pts_, k, d10)) {
                                    knn_window_group(pts_, Q4, P, k / NK, W, radius, zstep, ws->knn);
#pragma unroll
                                    for (int q = 0; q < 10; ++q) d10[q] = ws->knn[(size_t)k * 10 + q];
                                }
                                const int e0 = k * 10 - s0;
#pragma unroll
                                for (int q = 0; q < 10; ++q)
                                    if (e0 + q >= lo && e0 + q < hi) tile_[e0 + q - lo] = d10[q];
                            }
                            return;
                        }
                        if (grp >= ngroups) return;
                        for (int p = 0; p < NK; ++p) {
                            const int k = grp * NK + p;
                            const int e0 = e_first + p * 10 - s0;             // position inside this chunk
                            if (k >= P || e0 + 10 <= lo || e0 >= hi) continue;
                            double d10[10];
                            point_dists(pts_, grp, k, searched, d10);
#pragma unroll
                            for (int q = 0; q < 10; ++q)
                                if (e0 + q >= lo && e0 + q < hi) tile_[e0 + q - lo] = d10[q];
                        }
                    };
                    if (PW_IS_LDS(tile) && PW_IS_LDS(pts)) fill(PW_AS_LDS(tile), PW_AS_LDS(pts)); else fill(tile, pts);
                    T::sync();
                    np_leaf_phase<T>(tile, len, lo, hi, s_tab, s_acc, s_leaf, true);
                    lo = hi;
                }
                double part = np_walk_phase<T>(len, s_tab, s_acc, s_leaf);
                if (T::tid() == 0) total = first ? part : total + part;
                first = false;
                T::sync();
            }
            if (T::tid() == 0) v.red_v[15] = total;
            T::sync();
            sum = v.red_v[15];
            T::sync();
        } else {
            sum = np_sum_team<T>(ws->knn, P * 10, s_tab, s_acc, s_leaf, &v.red_v[15]);
        }
        double m = sum / (double)(P * 10);
        if (T::tid() == 0) { v.eps = m + pw_pow_np(m, 0.5); out->eps = v.eps; }
        T::sync();
        if (T::wave() == 0) PW_T1(ws, 25, t_sum);
    }
    if (T::wave() == 0) PW_T1(ws, 8, t_eps);
    PW_T0(t_smp);
    arena = arena_pts;
    double* vals = (double*)arena.take((size_t)P * 8);
    if (!vals) vals = ws->vals;
    int* surv_k = (int*)arena.take((size_t)P * 4);
    if (!surv_k) surv_k = ws->surv_k;
    int* labels = (int*)arena.take((size_t)P * 4);
    if (!labels) labels = ws->labels;
    unsigned char* flag = (unsigned char*)arena.take((size_t)P);
    if (!flag) flag = ws->flag;
    // everything taken below is temporary: the path-scan values of the sampling stage, then the
    // adjacency rows of DBSCAN occupy the same bytes in turn
    ScratchArena arena_mark = arena;
    double* tmpv = (double*)arena.take((size_t)P * 8);
    if (!tmpv) tmpv = ws->knn;            // (the k-NN rows are not needed once eps is known)
    // ---- sampling vectors (utilities.py:1457-1467): ray pre-analysis for every vector,
    //      then the coarse path scan for the vectors that hit nothing.  Two dense phases
    //      with an order-preserving compaction in between, so no lane idles while its
    //      neighbours walk a path.
    {
        double cen[3] = {v.centroid[0], v.centroid[1], v.centroid[2]};
        // atom by atom over the rays inside each atom's cone (team_ray_tests); the pair list borrows what is
        // left of the arena (the path-scan values are not alive yet), else the k-NN rows of the workspace
        bool done = false;
        {
            for (int k = T::tid(); k < P; k += T::SIZE) flag[k] = 1;
            ScratchArena a2 = arena_mark;
            ConeBand* bands = (ConeBand*)a2.take((size_t)n * sizeof(ConeBand));
            int cap = (int)(a2.left / 4);
            unsigned* pairs = (unsigned*)a2.take((size_t)cap * 4);
            if (cap < 4 * P) { pairs = (unsigned*)ws->knn; cap = 16 * ws->p_cap; }
            if (!bands && (size_t)n * sizeof(ConeBand) <= (size_t)ws->p_cap * 16) bands = (ConeBand*)(ws->knn + 8 * (size_t)ws->p_cap);
            auto run = [&](auto pts_) __attribute__((always_inline)) {
                SpiralRays<decltype(pts_)> getp{pts_, Q4};
                return team_ray_tests<T, false>(sh.S, n, cen, sp, getp, bands, pairs, cap, (PW_LDS int*)&v.red_i[0], flag, 0, nullptr);
            };
            if (bands) done = PW_IS_LDS(pts) ? run(PW_AS_LDS(pts)) : run(pts);
        }
        if (done) {
        } else if (T::SIZE > 1) {
            constexpr int NR = 4;
            for (int k0 = T::tid(); k0 < P; k0 += NR * T::SIZE) {
                double dx[NR], dy[NR], dz[NR], far[NR];
                bool hit[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    int k = k0 + r * T::SIZE < P ? k0 + r * T::SIZE : k0;
                    dx[r] = pts[PT(k, 0)]; dy[r] = pts[PT(k, 1)]; dz[r] = pts[PT(k, 2)];
                }
                ray_scan_multi<NR, false>(sh.S, n, cen, dx, dy, dz, hit, far);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    int k = k0 + r * T::SIZE;
                    if (k < P) flag[k] = hit[r] ? 0 : 1;
                }
            }
        } else {
            for (int k = T::tid(); k < P; k += T::SIZE) {
                double far;
                bool hit = ray_scan(sh.S, n, cen, pts[PT(k, 0)], pts[PT(k, 1)], pts[PT(k, 2)], &far);
                flag[k] = hit ? 0 : 1;
            }
        }
        T::sync();
        if (T::wave() == 0) {
            int m = 0;
            for (int base = 0; base < P; base += T::WSIZE) {
                int k = base + T::lane();
                bool f = k < P && flag[k] != 0;
                unsigned long long bal = T::ballot(f);
                int pos = m + __builtin_popcountll(bal & ((1ull << T::lane()) - 1ull));
                if (f) labels[pos] = k;          // labels[] is free until DBSCAN
                m += __builtin_popcountll(bal);
            }
            if (T::lane() == 0) v.n_surv = m;
        }
        T::sync();
        int ncand = v.n_surv;
        int evals = 0;
        PW_DCHECK(ncand >= 0 && ncand <= P, 104);
        if (T::wave() == 0) PW_T1(ws, 30, t_smp);     // rays + compaction
        PW_T0(t_path);
        // whole rounds: one path per thread.  The last, partial round would keep a handful of
        // lanes busy for a full path each, so its paths are cut into points instead: one
        // (path, point) per thread, then one thread per path folds its points.
        // (a last round that still fills a quarter of the team stays with one path per thread: a
        // thread's six points share every atom read, a (path, point) item reads all atoms for one)
        // (the first point of every path: the origin)
        const double m_origin = wave_gap_value<T>(sh.S, n, 0.0, 0.0, 0.0);
        int whole = T::SIZE > 1 ? (ncand / T::SIZE) * T::SIZE : ncand;
        // (a left-over path costs a wave 10 us, a partial round 40: from sixteen paths on the round is the cheaper)
        if ((ncand - whole) * 16 >= T::SIZE) whole = ncand;
        // (a wave's 64 paths -- consecutive survivors: a band of latitudes -- go through the atoms that can matter to
        // them, two thirds of the molecule: wave_path_candidates; the list lives behind the path values)
        lint* cand_w = nullptr;
        lint* coff_w = nullptr;
        {
            ScratchArena a3 = arena;
            const size_t per_wave = (size_t)n + PW_KCLS + 2;
            int* c_all = (int*)a3.take((size_t)T::NWAVES * per_wave * 4);
#ifndef PW_NO_SCAN_LISTS      // (tests/test_scan_lists.py builds the host probe with and without: the same records)
            if (c_all && PW_IS_LDS(c_all)) {
                cand_w = (lint*)PW_AS_LDS(c_all) + (size_t)T::wave() * per_wave;
                coff_w = cand_w + n;
            }
#endif
        }
        for (int j0 = 0; j0 < whole; j0 += T::SIZE) {
            const int j = j0 + T::tid();
            const bool have = j < whole;
            const int k = have ? labels[j] : 0;
            PW_DCHECK(k >= 0 && k < P, 105);
            const double vx = pts[PT(k, 0)], vy = pts[PT(k, 1)], vz = pts[PT(k, 2)];
            const bool listed = cand_w != nullptr &&
                                wave_path_candidates<T>(sh.S, n, have, vx, vy, vz, prm.increment, m_origin, cand_w, coff_w);
            if (have) {
                double g2, chunk[3];
                int pos;
                bool ok = path_scan_thread(sh.S, n, vx, vy, vz, prm.increment, m_origin, &g2, &pos, chunk, &evals,
                                           listed ? cand_w : nullptr, listed ? coff_w : nullptr);
                flag[j] = ok ? 1 : 0;
                tmpv[j] = g2;
            }
        }
        const int left = ncand - whole;
        if (left > 0) {
            // The paths of a last, partial round (fewer than a quarter of the team's threads): ONE WAVE per path, the
            // atoms spread over its lanes, point after point (wave_gap_value: the same value as the one-thread scan).
            // (Until round 6 a path's points went one to a thread, every thread through all atoms for its point: nine
            // lanes of a wave busy, 12.7 us for the four paths a CC3 frame usually leaves -- in-kernel timer.)
            for (int j = whole + T::wave(); j < ncand; j += T::NWAVES) {
                const int k = labels[j];
                const double vx = pts[PT(k, 0)], vy = pts[PT(k, 1)], vz = pts[PT(k, 2)];
                const int chunks = (int)np_floordiv(norm3(vx, vy, vz), prm.increment);
                double g2 = 0.0;
                bool ok = true;
                bool all_at_once = false;
                if constexpr (T::WSIZE == 64) all_at_once = chunks >= 1 && chunks <= 16 && sh.S.cls->k > 0;
                if (all_at_once) {
                    if constexpr (T::WSIZE == 64) {
                        // All points of the path at once: four lanes to a point, each a quarter of every radius group
                        // (points_gap_values' arithmetic: the smallest squared distance of a group, then one root), the
                        // four folded by lane exchanges; then the path's statements over the points in order.
                        const auto& C = *sh.S.cls;
                        const Frame& F = sh.S;
                        const double cx = vx / (double)chunks, cy = vy / (double)chunks, cz = vz / (double)chunks;
                        const int sub = T::lane() & 3;
                        const int q = (T::lane() >> 2) + 1 <= chunks ? (T::lane() >> 2) + 1 : chunks;
                        const double qx = cx * (double)q, qy = cy * (double)q, qz = cz * (double)q;
                        const double pp = sq3(qx, qy, qz);
                        double gap = PW_INF;
                        for (int g = 0; g < C.k; ++g) {
                            double m2 = PW_INF;
                            const int hi = C.off[g + 1];
                            for (int i = C.off[g] + sub; i < hi; i += 4) {
                                const double x = F.x[i], y = F.y[i], z = F.z[i], xx = F.xx[i];
                                const double gg = pw_fma(z, qz, pw_fma(x, qx, y * qy));
                                m2 = __builtin_fmin(m2, pw_m2add(gg, xx));
                            }
                            m2 = __builtin_fmin(m2, T::xor_d(m2, 1));
                            m2 = __builtin_fmin(m2, T::xor_d(m2, 2));
                            const double m2p = m2 + pp;
                            const double d = pw_sqrt(m2p > 0.0 ? m2p : 0.0);
                            gap = __builtin_fmin(gap, d - C.vdw[g]);
                        }
                        double best = m_origin;
                        if (!(m_origin > 0.0)) ok = false;
                        for (int q2 = 1; q2 <= chunks && ok; ++q2) {
                            const double m = T::bcast_u(gap, 4 * (q2 - 1));
                            if (!(m > 0.0)) ok = false;
                            else if (m < best) best = m;
                        }
                        g2 = best * 2.0;
                    }
                } else if (chunks >= 1) {
                    // (path_scan_thread's statements: the first point is the origin, its gap the caller's m_origin)
                    const double cx = vx / (double)chunks, cy = vy / (double)chunks, cz = vz / (double)chunks;
                    double best = m_origin;
                    if (!(m_origin > 0.0)) ok = false;
                    for (int q = 1; q <= chunks && ok; ++q) {
                        const double m = wave_gap_value<T>(sh.S, n, cx * (double)q, cy * (double)q, cz * (double)q);
                        if (!(m > 0.0)) ok = false;
                        else if (m < best) best = m;
                    }
                    g2 = best * 2.0;
                } else {
                    double chunk[3];
                    int pos;
                    ok = path_scan_thread(sh.S, n, vx, vy, vz, prm.increment, m_origin, &g2, &pos, chunk, &evals);
                }
                if (T::lane() == 0) { flag[j] = ok ? 1 : 0; tmpv[j] = g2; }
            }
        }
        T::sync();
        if (T::wave() == 0) PW_T1(ws, 31, t_path);    // path scans
        if (T::wave() == 0) {
            int m = 0;
            for (int base = 0; base < ncand; base += T::WSIZE) {
                int j = base + T::lane();
                bool f = j < ncand && flag[j] != 0;
                unsigned long long bal = T::ballot(f);
                int pos = m + __builtin_popcountll(bal & ((1ull << T::lane()) - 1ull));
                if (f) { surv_k[pos] = labels[j]; vals[pos] = tmpv[j]; }
                m += __builtin_popcountll(bal);
            }
            if (T::lane() == 0) { v.n_surv = m; out->n_survivors = m; }
        }
        T::sync();
        (void)evals;
    }
    if (T::wave() == 0) PW_T1(ws, 9, t_smp);
    int ns = v.n_surv;
    if (ns == 0) {
        // no vector reaches the outside: find_windows returns None
        if (T::tid() == 0) out->n_windows = -1;
        T::sync();
        return -1;
    }
    PW_T0(t_db);
    arena = arena_mark;                   // tmpv is dead: its values were compacted into vals
    // ---- DBSCAN(eps, min_samples = 5) on the survivors' end points -----------------------
    {
        const int label = team_dbscan<T>(sh.bits[0], sh.bits[1], sh.bits[2], arena, ws, pts, PT, surv_k, ns, v.eps, labels);
        if (label < 0) {
            if (T::tid() == 0) v.status |= PW_ST_POINTS_OVERFLOW;
            T::sync();
            return -1;
        }
        if (T::tid() == 0) {
            v.n_clusters = label;
            out->n_clusters = label;
        }
        // the per-cluster arrays: team LDS for what a record holds, the team's global slab beyond
        // (clusters <= core points <= survivors <= p_cap)
        if (label > PW_W_MAX) {
            wa.vec = ws->xw; wa.d = ws->xw + 3 * (size_t)ws->p_cap; wa.c = ws->xw + 4 * (size_t)ws->p_cap; wa.ok = ws->xw_ok;
        }
        for (int c = T::tid(); c < label; c += T::SIZE) wa.ok[c] = 0;
        T::sync();
        // utilities.py:1221 -- per cluster the vector with the largest 2*gap (first occurrence)
        for (int c = T::wave(); c < label; c += T::NWAVES) {
            double best = -PW_INF;
            int bidx = 0x7fffffff;
            for (int q = T::lane(); q < ns; q += T::WSIZE) {
                if (labels[q] == c) {
                    double val = vals[q];
                    if (val > best || (val == best && q < bidx)) { best = val; bidx = q; }
                }
            }
            T::wave_argmax(best, bidx);
            if (T::lane() == 0) {
                const int pvk = surv_k[bidx];
                wa.vec[3 * c] = pts[PT(pvk, 0)]; wa.vec[3 * c + 1] = pts[PT(pvk, 1)]; wa.vec[3 * c + 2] = pts[PT(pvk, 2)];
            }
        }
        T::sync();
        // stage capture (pw_analysis_debug): survivors in pass order, their labels and path minima
        if (pw_unit_debug* dbg = ws->dbg_base ? ws->dbg_base + ws->unit : nullptr) {
            for (int i = T::tid(); i < ns && i < PW_P_MAX; i += T::SIZE) {
                dbg->pass_idx[i] = surv_k[i];
                dbg->labels[i] = labels[i];
                dbg->gap2[i] = vals[i];
            }
            if (T::tid() == 0) { dbg->n_survivors = ns; dbg->n_clusters = label; }
        }
    }
    if (T::wave() == 0) PW_T1(ws, 10, t_db);
    return v.n_clusters;
}

// result assembly (utilities.py:1526-1536) by ONE thread: the fitted windows in cluster order; the record holds
// PW_W_MAX of them, the others go to the launch's extra-window list.  Returns the status bits to merge.
PW_HD inline int windows_finish(const WinArrays& wa, int ncl, pw_unit_out* out, TeamWorkspace* ws, long unit) {
    int st = 0, m = 0;
    for (int c = 0; c < ncl; ++c) {
        if (wa.ok[c] > 0) {
            const double wd = wa.d[c];
            if (m < PW_W_MAX) {
                out->win_d[m] = wd;
                out->win_c[m][0] = wa.c[3 * c];
                out->win_c[m][1] = wa.c[3 * c + 1];
                out->win_c[m][2] = wa.c[3 * c + 2];
            } else {
                st |= PW_ST_WINDOW_OVERFLOW;
                unsigned slot = team_atomic_inc(ws->xwin_count);
                if (ws->xwin && slot < ws->xwin_cap) {
                    pw_extra_window* e = ws->xwin + slot;
                    e->unit = unit; e->index = m; e->reserved = 0; e->d = wd;
                    e->c[0] = wa.c[3 * c]; e->c[1] = wa.c[3 * c + 1]; e->c[2] = wa.c[3 * c + 2];
                }
            }
            if (wd < 0.0) st |= PW_ST_WINDOW_NEGATIVE;
            ++m;
        } else if (wa.ok[c] < 0) {
            st |= PW_ST_Z_BOUNDS;
        } else {
            st |= PW_ST_WINDOW_DROPPED;
        }
    }
    out->n_windows = m;
    return st;
}

template <class T>
PW_HD inline __attribute__((always_inline)) void stage_windows_impl(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                                                    const pw_params& prm) {
    auto& v = *sh.v;
    WinArrays wa;
    wa.vec = (double*)&v.win_vec[0][0]; wa.d = (double*)v.win_d; wa.c = (double*)&v.win_c[0][0]; wa.ok = (int*)v.win_ok;
    if (windows_bulk_impl<T>(sh, ws, n, out, prm, wa) < 0) return;
    PW_T0(t_w);
    // ---- one window per cluster, clusters dealt round-robin to the waves -------------------
    const int ncl = v.n_clusters;
    // at most four waves fit windows at a time (one rotated frame + optimiser state each);
    // in an 8-wave team the upper four only take part in the bulk stages
    int nslot = T::NWAVES < 4 ? T::NWAVES : 4;
    if (sh.nslots < nslot) nslot = sh.nslots;       // (a launch may carve fewer slots than waves to save LDS)
    if (T::wave() < nslot) {
        const double shift[3] = {v.shift[0], v.shift[1], v.shift[2]};
        for (int c = T::wave(); c < ncl; c += nslot)
            wave_window<T>(sh.S, sh.R[T::wave()], sh.lb[T::wave()], ws, n, c, shift, prm, wa, (int*)&v.red_i[8 + T::wave()]);
    }
    T::sync();
    if (T::wave() == 0) PW_T1(ws, 12, t_w);
    if (T::tid() == 0) {
        v.status |= windows_finish(wa, ncl, out, ws, ws->unit);
        for (int w = 0; w < T::NWAVES && w < 8; ++w) v.n_eval += v.red_i[8 + w];
    }
    T::sync();
}

// out of line for the kernels that hold several stages; the window launch's own kernel (PW_KERNEL_WINDOWS) inlines the
// stage -- a kernel saves no callee-saved registers, an out-of-line stage that fills the register file saves a hundred
template <class T>
PW_NOINLINE PW_HD inline void stage_windows(UnitShared& sh, TeamWorkspace* ws, int n, pw_unit_out* out,
                                             const pw_params& prm) {
    stage_windows_impl<T>(sh, ws, n, out, prm);
}

// ---- the unit ------------------------------------------------------------------------------------
// internal stage bits used when one analysis is split over several launches
constexpr unsigned PW_STAGE_REUSE_OPT = 16u;   // pore centre already in the record (earlier launch)
constexpr unsigned PW_STAGE_MERGE = 32u;       // record is shared with other launches: no resets
constexpr unsigned PW_STAGE_COM_ONLY = 64u;    // only what later stages need from stage_basic

PW_HD inline void record_or_status(pw_unit_out* out, int st, int evals) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (st) atomicOr(&out->status, st);
    atomicAdd(&out->n_eval, evals);
#else
    out->status |= st;
    out->n_eval += evals;
#endif
}

constexpr unsigned PW_KERNEL_AVERAGE = PW_STAGE_AVG | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;
constexpr unsigned PW_KERNEL_WINDOWS = PW_STAGE_WINDOWS | PW_STAGE_REUSE_OPT | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;

template <class T, unsigned KMASK = 0xffffffffu>
PW_HD inline void analyse_unit(UnitShared& sh, TeamWorkspace* ws, int n, const double* xyz,
                               const double* vdw, const double* mass, unsigned stages,
                               pw_unit_out* out, const pw_params& prm, const unsigned char* tmpl = nullptr) {
    const bool merge = (stages & PW_STAGE_MERGE) != 0;
    const bool reuse_opt = (stages & PW_STAGE_REUSE_OPT) != 0;
    if ((stages & PW_STAGE_WINDOWS) && !reuse_opt && prm.pore_opt) stages |= PW_STAGE_OPT;
    if (T::tid() == 0 && !merge) {
        out->status = 0;
        out->n_eval = 0;
        out->avg_d = 0.0;
        out->pore_opt_d = 0.0; out->pore_opt_atom = -1; out->pore_vol_opt = 0.0;
        out->pore_opt_c[0] = out->pore_opt_c[1] = out->pore_opt_c[2] = 0.0;
        out->n_windows = -1; out->n_clusters = 0;
        out->n_points = 0; out->n_points_avg = 0; out->n_survivors = 0;
        out->opt_nit = 0; out->opt_nfev = 0; out->opt_task = 0; out->opt_msg = 0;
        out->eps = 0.0; out->sphere_r = 0.0;
        for (int w = 0; w < PW_W_MAX; ++w) {
            out->win_d[w] = 0.0;
            out->win_c[w][0] = out->win_c[w][1] = out->win_c[w][2] = 0.0;
        }
    }
    PW_T0(t_load);
    load_unit<T>(sh, n, xyz, vdw, mass, tmpl);
#ifndef PW_TIME_BASIC
    if (T::wave() == 0) PW_T1(ws, 5, t_load);      // (diagnostic builds: slot 5, every launch's load stage together)
#endif
    if (reuse_opt) {
        // the optimiser launch already wrote the centre of mass
        if (T::tid() == 0) { sh.v->com[0] = out->com[0]; sh.v->com[1] = out->com[1]; sh.v->com[2] = out->com[2]; }
        T::sync();
    } else if (KMASK == PW_KERNEL_AVERAGE) {
        stage_basic_impl<T>(sh, ws, n, out, true);
    } else {
#ifdef PW_TIME_BASIC
        PW_T0(t_basic);
#endif
        stage_basic<T>(sh, ws, n, out, (stages & PW_STAGE_COM_ONLY) != 0);
#ifdef PW_TIME_BASIC
        if (T::wave() == 0) PW_T1(ws, 5, t_basic);       // (diagnostic: slot 5 is then the basic stage, not the load)
#endif
    }
    if (stages & PW_STAGE_OPT) stage_opt<T>(sh, ws, n, out, prm);
    if (reuse_opt) {
        if (T::tid() == 0) {
            sh.v->opt_c[0] = out->pore_opt_c[0];
            sh.v->opt_c[1] = out->pore_opt_c[1];
            sh.v->opt_c[2] = out->pore_opt_c[2];
            if (out->status & PW_ST_NEGATIVE_PORE) sh.v->status |= PW_ST_NEGATIVE_PORE;
        }
        T::sync();
    }
    if (stages & PW_STAGE_AVG) {
        PW_T0(t_a);
        if (KMASK == PW_KERNEL_AVERAGE) stage_average_impl<T, true>(sh, ws, n, out, prm);
        else stage_average<T>(sh, ws, n, out, prm);
        if (T::wave() == 0) PW_T1(ws, 13, t_a);
        if ((stages & PW_STAGE_WINDOWS) && sh.S.x == sh.A.x) {
            // A team that keeps ONE frame has just shifted it in place (to the centre of mass) and the window search
            // shifts the INPUT frame (to the pore centre): the coordinates come in again -- the scatter of load_unit,
            // the grouping is kept -- instead of every window team of the pipeline carrying a second frame (5 KB of LDS)
            for (int i = T::tid(); i < n; i += T::SIZE) {
                const int pos = sh.inv[i];
                const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
                sh.A.x[pos] = x; sh.A.y[pos] = y; sh.A.z[pos] = z;
                sh.A.xx[pos] = sq3(x, y, z);
            }
            T::sync();
        }
    }
    if (stages & PW_STAGE_WINDOWS) {
        if (!(prm.pore_opt && (sh.v->status & PW_ST_NEGATIVE_PORE))) {
#ifdef PW_INLINE_WINDOW_STAGE
            if (KMASK == PW_KERNEL_WINDOWS) stage_windows_impl<T>(sh, ws, n, out, prm);
            else
#endif
            stage_windows<T>(sh, ws, n, out, prm);
        }
        else if (T::tid() == 0) out->n_windows = -1;     // no window search: None, whichever launch shape
    }
    if (T::tid() == 0) {
        if (merge) {
            record_or_status(out, sh.v->status, sh.v->n_eval);
        } else {
            out->status = sh.v->status;
            out->n_eval = sh.v->n_eval;
        }
    }
    T::sync();
}

}  // namespace pw
