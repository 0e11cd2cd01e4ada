// tests/hostsim/unit_probe.cpp -- runs the unit pipeline of
// pywindow_amd/csrc/pw_unit.hpp on the HOST with a one-thread team, so the
// control flow of the HIP kernels can be checked against the golden vectors in
// a container without a GPU.  Test infrastructure only: the product library
// (libpywindow_hip.so) never links or calls this.
#include <vector>
#include "../../pywindow_amd/csrc/pw_unit.hpp"
#include <stdlib.h>
#include <string.h>
using namespace pw;
static int run_batch(long n_units, const long* off, const double* xyz, const double* vdw, const double* mass,
                     unsigned stages, pw_unit_out* out, const pw_params* params, pw_unit_debug* dbg,
                     int p_cap = 0, pw_extra_window* xw = nullptr, unsigned xw_cap = 0, unsigned* xw_count = nullptr);
// the same with an explicit sampling-vector capacity (0: from the adjust knobs, as the library sizes it) and
// the list of windows beyond PW_W_MAX
extern "C" int hs_analysis_ext(long n_units, const long* off, const double* xyz, const double* vdw,
                               const double* mass, unsigned stages, pw_unit_out* out, const pw_params* params,
                               int p_cap, pw_extra_window* xw, unsigned xw_cap, unsigned* xw_count) {
    return run_batch(n_units, off, xyz, vdw, mass, stages, out, params, nullptr, p_cap, xw, xw_cap, xw_count);
}
extern "C" int hs_analysis_batch(long n_units, const long* off, const double* xyz, const double* vdw,
                                 const double* mass, unsigned stages, pw_unit_out* out,
                                 const pw_params* params) {
    return run_batch(n_units, off, xyz, vdw, mass, stages, out, params, nullptr);
}
// the same with the stage capture of find_windows (pw_unit_debug, one record per unit)
extern "C" int hs_analysis_debug(long n_units, const long* off, const double* xyz, const double* vdw,
                                 const double* mass, unsigned stages, pw_unit_out* out, pw_unit_debug* dbg) {
    memset(dbg, 0, sizeof(pw_unit_debug) * (size_t)n_units);
    return run_batch(n_units, off, xyz, vdw, mass, stages, out, nullptr, dbg);
}
extern "C" int hs_sizeof_unit_debug() { return (int)sizeof(pw_unit_debug); }
static int run_batch(long n_units, const long* off, const double* xyz, const double* vdw, const double* mass,
                     unsigned stages, pw_unit_out* out, const pw_params* params, pw_unit_debug* dbg,
                     int p_cap, pw_extra_window* xw, unsigned xw_cap, unsigned* xw_count) {
    pw_params prm = default_params();
    if (params) prm = *params;
    if (p_cap <= 0) p_cap = params_p_cap(prm.adjust_windows, prm.adjust_average);
    p_cap = round_p_cap(p_cap);
    unsigned dummy_count = 0;
    if (!xw_count) xw_count = &dummy_count;
    *xw_count = 0;
    int nmax = 0;
    for (long u = 0; u < n_units; ++u) { int n = (int)(off[u + 1] - off[u]); if (n > nmax) nmax = n; }
    // (HS_NLB: optimiser-state slots carved = size of the team's scratch arena; 8 lets every LDS-resident
    // variant of the stages run on the host as it does in the kernels, 1 forces their fallbacks)
    const char* nlb_env = getenv("HS_NLB");
    const int nlb = nlb_env ? atoi(nlb_env) : 8;
    size_t bytes = UnitShared::bytes(nmax, 1, nlb, 2, false, p_cap);
    unsigned char* lds = (unsigned char*)aligned_alloc(16, (bytes + 15) & ~(size_t)15);
    TeamWorkspace* ws = (TeamWorkspace*)calloc(1, sizeof(TeamWorkspace));
    unsigned char* slab = (unsigned char*)malloc(team_slab_bytes(p_cap));
    if (!lds || !ws || !slab) return -5;
    bind_team_slab(ws, slab, p_cap);
    ws->adj = (unsigned long long*)malloc(sizeof(unsigned long long) * team_adj_words(p_cap));
    ws->xwin = xw; ws->xwin_cap = xw_cap; ws->xwin_count = xw_count;
    // neighbour tables of the sampling sphere (HS_NB_TABLES=0: without, the windowed search only): built for
    // the vector counts this batch needs -- found by a first pass without tables
    static std::vector<unsigned> nb_off(PW_NB_PMAX + 1, PW_NB_NONE);
    static std::vector<unsigned short> nb_idx;
    static std::vector<double> nb_bound;
    const char* nbt = getenv("HS_NB_TABLES");
    const bool want_tables = !(nbt && nbt[0] == '0') && (stages & PW_STAGE_WINDOWS);
    ws->nb_off = nullptr; ws->nb_idx = nullptr; ws->nb_bound = nullptr;
    for (int pass = want_tables ? 0 : 1; pass < 2; ++pass) {
    if (pass == 1 && want_tables) {
        for (long u = 0; u < n_units; ++u) {
            const int P = out[u].n_points;
            if (P < PW_NB_PMIN || P > PW_NB_PMAX || nb_off[P] != PW_NB_NONE) continue;
            const unsigned first = (unsigned)nb_bound.size();
            std::vector<double> ux(P), uy(P), uz(P);
            Sphere sp;
            sp.init(1.0, P);
            for (int k = 0; k < P; ++k) sp.point(k, &ux[k], &uy[k], &uz[k]);
            nb_idx.resize((size_t)(first + P) * PW_NB_K);
            nb_bound.resize(first + P);
            for (int k = 0; k < P; ++k)
                nb_build_point(P, k, ux.data(), uy.data(), uz.data(), nb_idx.data() + (size_t)(first + k) * PW_NB_K, nb_bound.data() + first + k);
            nb_off[P] = first;
        }
        ws->nb_off = nb_off.data(); ws->nb_idx = nb_idx.data(); ws->nb_bound = nb_bound.data();
        *xw_count = 0;
    }
    static unsigned rsq_tab[65536];
    static bool rsq_ready = false;
    if (!rsq_ready) { rsqrt14_decode(rsq_tab); rsq_ready = true; }
    ws->rsq = rsq_tab;
    ws->dbg_base = dbg;
    // HS_LDS_FILL=<byte>: what team-shared memory holds before a unit starts (on the GPU: whatever the
    // previous workgroup left).  Results must not depend on it.
    const char* fill_env = getenv("HS_LDS_FILL");
    const int fill = fill_env ? (int)strtol(fill_env, nullptr, 0) : 0xff;    // NaNs / -1 by default
    const char* fa_ = getenv("HS_LDS_FILL_FROM");
    const char* fb_ = getenv("HS_LDS_FILL_TO");
    for (long u = 0; u < n_units; ++u) {
        ws->unit = u;
        memset(lds, (fa_ || fb_) ? 0 : fill, bytes);
        {
            // HS_LDS_FILL_FROM / HS_LDS_FILL_TO: poison only that byte range (to locate a dependence)
            const char* fa = getenv("HS_LDS_FILL_FROM");
            const char* fb = getenv("HS_LDS_FILL_TO");
            size_t a = fa ? (size_t)strtol(fa, nullptr, 0) : 0, b = fb ? (size_t)strtol(fb, nullptr, 0) : bytes;
            if (b > bytes) b = bytes;
            if ((fa || fb) && a < b) memset(lds + a, fill, b - a);
        }
        UnitShared sh;
        sh.carve(lds, nmax, 1, nlb, 2, false, p_cap);
        int n = (int)(off[u + 1] - off[u]);
        memset(&out[u], 0, sizeof(pw_unit_out));
        analyse_unit<HostTeam>(sh, ws, n, xyz + 3 * off[u], vdw + off[u], mass + off[u], stages, &out[u], prm);
    }
    }
    free(ws->adj); free(slab); free(lds); free(ws);
    return 0;
}
extern "C" int hs_sizeof_unit_out() { return (int)sizeof(pw_unit_out); }
// DBSCAN(eps, min_samples = 5) as the one-thread team computes it (the same source as pw_dbscan); points n x 3
extern "C" int hs_dbscan(const double* points, long n, double eps, int* labels) {
    static unsigned long long bits[3][PW_DBSCAN_MAX / 64];
    if (n > PW_DBSCAN_MAX) return -2;
    std::vector<double> soa((size_t)3 * n);
    std::vector<int> ident((size_t)n);
    for (long i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) soa[(size_t)k * n + i] = points[3 * i + k];
        ident[i] = (int)i;
    }
    TeamWorkspace* ws = (TeamWorkspace*)calloc(1, sizeof(TeamWorkspace));
    const int p_cap = round_p_cap(n);
    std::vector<unsigned long long> adj(team_adj_words(p_cap));
    ws->adj = adj.data();
    ws->p_cap = p_cap;
    ScratchArena arena;
    arena.cur = nullptr;
    arena.left = 0;
    const int nn = (int)n;
    auto PT = [nn](int k, int c) { return c * nn + k; };
    int k = team_dbscan<HostTeam>(bits[0], bits[1], bits[2], arena, ws, soa.data(), PT, ident.data(), nn, eps, labels);
    free(ws);
    return k;
}
// numpy's add.reduce order as the one-thread team computes it (the same source as pw_pairwise_sum)
extern "C" double hs_pairwise_sum(const double* a, long n) {
    static int tab[324];
    static double acc[8 * 160], leaf[256];
    double slot = 0.0;
    return np_sum_team<HostTeam>(a, (int)n, tab, acc, leaf, &slot);
}
// ... and the serial restatement it replaced in round 1 (np_sum_serial), kept as a cross-check
extern "C" double hs_pairwise_sum_serial(const double* a, long n) { return np_sum_serial(a, (int)n); }
// ... and the scalar-state variant the periodic re-assembly uses (np_sum_lean)
extern "C" double hs_pairwise_sum_lean(const double* a, long n) { return np_sum_lean(a, (int)n); }
extern "C" long hs_lds_bytes(int nmax) { return (long)UnitShared::bytes(nmax, 1, 8); }
extern "C" long hs_lds_offset(int nmax, int what) {
    // byte offsets of the parts of the team-shared block (for the poison tests)
    static unsigned char dummy[1];
    UnitShared sh;
    sh.carve(dummy, nmax, 1, 8);
    const unsigned char* base = dummy;
    switch (what) {
        case 0: return (long)((const unsigned char*)sh.v - base);
        case 1: return (long)((const unsigned char*)sh.vdw - base);
        case 2: return (long)((const unsigned char*)sh.mass - base);
        case 3: return (long)((const unsigned char*)sh.perm - base);
        case 4: return (long)((const unsigned char*)sh.inv - base);
        case 5: return (long)((const unsigned char*)sh.A.x - base);
        case 6: return (long)((const unsigned char*)sh.S.x - base);
        case 7: return (long)((const unsigned char*)sh.R[0].x - base);
        case 8: return (long)((const unsigned char*)sh.lb[0] - base);
        default: return (long)UnitShared::bytes(nmax, 1, 8);
    }
}
