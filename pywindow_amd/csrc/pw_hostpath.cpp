// pw_hostpath.cpp -- the `device = -1` path of the C ABI (include/pywindow_amd.h): the SAME unit pipeline
// (pw_unit.hpp, single source with the gfx950 kernels) compiled by g++ for a one-lane team and run by host
// threads over the units of a batch.  Explicit choice only -- pw_context_create(-1); nothing ever falls back
// to it.  It is what makes BASELINE.json's configs[0] ("CC3 single-frame full_analysis() on CPU") runnable
// through the product's own boundary, and what bench.py times as the same-source CPU figure.
//
// Reference for the path: Molecule.full_analysis (molecular.py:156-202) per unit, the per-frame loop of
// Trajectory._analysis_serial (trajectory.py:496-522) over the batch; `threads` plays the role of
// analysis(ncpus=...) (trajectory.py:553-586).
//
// Built with g++ -O2 -ffp-contract=off -mfma (fused multiply-adds only where the source writes them).
#define pw pw_cpu          // a namespace of its own: nothing here merges with the HIP translation units' host code
#include "pw_unit.hpp"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

using namespace pw;

namespace {

// neighbour tables of the sampling sphere (pw_unit.hpp), built per vector count on first use
struct HostTables {
    std::mutex lock;
    // bumped (release) after a table has been appended: a worker copies `off` under the lock when the version
    // it last saw is stale, and reads rows only through its copy -- every row it can reach was complete before
    // the version it acquired (no plain read of memory another thread may be writing)
    std::atomic<unsigned> version{0};
    std::vector<unsigned> off = std::vector<unsigned>(PW_NB_PMAX + 1, PW_NB_NONE);
    // one allocation per P (pointers handed to running threads must stay valid): offsets index `rows`
    std::vector<unsigned short> idx;
    std::vector<double> bound;
    HostTables() {
        // room for every P up front would be 84 MB; tables are appended instead and the vectors reserved
        // generously so that appending never moves them while other threads read
        idx.reserve((size_t)64 * PW_NB_PMAX * PW_NB_K);
        bound.reserve((size_t)64 * PW_NB_PMAX);
    }
    void ensure(int P) {
        if (P < PW_NB_PMIN || P > PW_NB_PMAX) return;
        std::lock_guard<std::mutex> g(lock);
        if (off[P] != PW_NB_NONE) return;
        const size_t first = bound.size();
        if (first + (size_t)P > bound.capacity()) return;          // (64 distinct counts seen: no more tables)
        std::vector<double> ux(P), uy(P), uz(P);
        Sphere sp;
        sp.init(1.0, P);
        for (int k = 0; k < P; ++k) sp.point(k, &ux[k], &uy[k], &uz[k]);
        idx.resize((first + P) * PW_NB_K);
        bound.resize(first + P);
        for (int k = 0; k < P; ++k)
            nb_build_point(P, k, ux.data(), uy.data(), uz.data(), idx.data() + (first + k) * PW_NB_K, bound.data() + first + k);
        off[P] = (unsigned)first;
        version.fetch_add(1, std::memory_order_release);
    }
    // the worker's private view of `off` (see `version`)
    void snapshot(std::vector<unsigned>& mine, unsigned& seen) {
        const unsigned v = version.load(std::memory_order_acquire);
        if (v == seen && !mine.empty()) return;
        std::lock_guard<std::mutex> g(lock);
        mine = off;
        seen = version.load(std::memory_order_relaxed);
    }
};
HostTables g_tables;
unsigned g_rsq[65536];
std::once_flag g_rsq_once;

}  // namespace

// One batch on the host.  Returns 0 or PW_E_NOMEM.  extra: windows beyond PW_W_MAX, appended unsorted.
extern "C" int pw_hostpath_run(const pw_batch_in* in, unsigned stages, pw_unit_out* out, const pw_params* prm_in,
                               int p_cap, int threads, pw_unit_debug* dbg, pw_extra_window* xw, unsigned xw_cap,
                               unsigned* xw_count) {
    std::call_once(g_rsq_once, [] { rsqrt14_decode(g_rsq); });
    const pw_params prm = prm_in ? *prm_in : default_params();
    const long n_units = (long)in->n_units;
    int nmax = 0;
    for (long u = 0; u < n_units; ++u) nmax = std::max(nmax, (int)(in->atom_offset[u + 1] - in->atom_offset[u]));
    p_cap = round_p_cap(p_cap);
    if (threads < 1) threads = 1;
    if ((long)threads > n_units) threads = (int)std::max(1l, n_units);
    std::atomic<long> next{0};
    std::atomic<int> failed{0};
    unsigned xcount = 0;            // (bumped with atomic increments: pw_unit.hpp team_atomic_inc)
    const int vstride = in->template_atoms > 0 ? 0 : 1;
    auto worker = [&]() {
        const size_t bytes = UnitShared::bytes(nmax, 1, 8, 2, false, p_cap);
        unsigned char* lds = (unsigned char*)aligned_alloc(16, (bytes + 15) & ~(size_t)15);
        TeamWorkspace* ws = (TeamWorkspace*)calloc(1, sizeof(TeamWorkspace));
        unsigned char* slab = (unsigned char*)malloc(team_slab_bytes(p_cap));
        unsigned long long* adj = (unsigned long long*)malloc(sizeof(unsigned long long) * team_adj_words(p_cap));
        if (!lds || !ws || !slab || !adj) {
            failed = 1;
            free(lds); free(ws); free(slab); free(adj);
            return;
        }
        bind_team_slab(ws, slab, p_cap);
        ws->adj = adj;
        ws->rsq = g_rsq;
        ws->dbg_base = dbg;
        ws->xwin = xw; ws->xwin_cap = xw_cap; ws->xwin_count = &xcount;
        std::vector<unsigned> nb_off;
        unsigned nb_seen = 0;
        for (;;) {
            const long u = next.fetch_add(1);
            if (u >= n_units) break;
            ws->unit = u;
            g_tables.snapshot(nb_off, nb_seen);
            ws->nb_off = nb_off.data(); ws->nb_idx = g_tables.idx.data(); ws->nb_bound = g_tables.bound.data();
            memset(lds, 0, bytes);
            UnitShared sh;
            sh.carve(lds, nmax, 1, 8, 2, false, p_cap);
            const long a0 = (long)in->atom_offset[u];
            const int n = (int)(in->atom_offset[u + 1] - a0);
            memset(&out[u], 0, sizeof(pw_unit_out));
            analyse_unit<HostTeam>(sh, ws, n, in->xyz + 3 * a0, in->vdw + a0 * vstride, in->mass + a0 * vstride, stages,
                                   &out[u], prm);

            if ((stages & PW_STAGE_WINDOWS) && out[u].n_points >= PW_NB_PMIN) g_tables.ensure(out[u].n_points);
        }
        free(adj); free(slab); free(ws); free(lds);
    };
    if (threads == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    if (xw_count) *xw_count = xcount;
    return failed ? PW_E_NOMEM : PW_OK;
}

extern "C" int pw_hostpath_default_threads(void) {
    const char* e = getenv("PW_CPU_THREADS");
    if (e && atoi(e) > 0) return atoi(e);
    unsigned hc = std::thread::hardware_concurrency();
    return hc ? (int)hc : 1;
}
