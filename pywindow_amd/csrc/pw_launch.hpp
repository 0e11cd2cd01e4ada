// pw_launch.hpp -- what a launch hands its teams, shared by the translation units that hold analysis
// kernels (pw_kernels.hip: team memory in LDS; pw_kernels_big.hip: team memory in global memory).
#pragma once
#include "../../include/pywindow_amd.h"

// Where a launch's teams find their global workspaces: headers, the per-team slabs of the per-vector
// arrays (sized by the launch's sampling-vector capacity p_cap), DBSCAN adjacency rows, the launch-wide
// list of windows beyond what a record holds, and the neighbour tables of the sampling sphere.
struct PwWsArgs {
    void* ws;                       // TeamWorkspace array of the launch
    unsigned char* slab;            // team t: slab + t * team_slab_bytes(p_cap)
    unsigned long long* adj;        // team t: adj + t * team_adj_words(p_cap); null: the launch runs no DBSCAN
    pw_extra_window* xwin;
    unsigned* xwin_count;
    unsigned xwin_cap;
    int p_cap;
    const unsigned* nb_off;         // neighbour tables of the sampling sphere (pw_unit.hpp), null: none
    const unsigned short* nb_idx;
    const double* nb_bound;
    const double* nb_unit;          // the unit vectors themselves, 3 per point (null: the teams compute them)
    // How long a team of one launch waits for ANOTHER launch of the same analysis without seeing it make any
    // progress (ticks of the 100 MHz wall clock; pw_context::wait_ticks, PW_WAIT_LIMIT_MS, default 250 ms) before it
    // gives the analysis up (PW_E_TIMEOUT).  A wait during which the other side keeps publishing is never cut short.
    long long wait_ticks;
    long long stream_wait_ticks;    // ... and for the HOST to append coordinates to a streamed batch (PW_STREAM_LIMIT_MS, default 5 s)
};

// Hand-off between the optimiser launch (producer, one wave per unit) and the window
// launch (consumer, persistent teams): a producer publishes the index of a unit whose pore
// centre is in its result record, consumers take published units in completion order.
//   producer: record stores -> s_waitcnt -> agent release fence -> s_waitcnt -> relaxed store
//   consumer: ONE relaxed poll loop -> agent acquire fence -> s_waitcnt -> team barrier -> loads
// (MI355X guide, "Inter-workgroup communication").  Every spin is bounded.
struct UnitQueue {
    unsigned long long tail;   // next free slot (producers)
    unsigned long long head;   // next slot to consume
    int error;                 // set when a consumer gives up waiting
    int started;               // producer teams that have begun (gate for the other launches)
};
enum : int { PW_ROLE_PLAIN = 0, PW_ROLE_PRODUCER = 1, PW_ROLE_CONSUMER = 2 };

#if defined(__HIPCC__)
// (slab_bytes / adj_words: team_slab_bytes(p_cap) / team_adj_words(p_cap) of the caller's pw_unit.hpp)
template <class WS>
__device__ inline void bind_workspace(WS* ws, const PwWsArgs& a, unsigned team, const unsigned* rsq_tab, size_t slab_bytes,
                                      size_t adj_words) {
    bind_team_slab(ws, a.slab + (size_t)team * slab_bytes, a.p_cap);
    ws->adj = a.adj ? a.adj + (size_t)team * adj_words : nullptr;
    ws->xwin = a.xwin; ws->xwin_count = a.xwin_count; ws->xwin_cap = a.xwin_cap;
    ws->nb_off = a.nb_off; ws->nb_idx = a.nb_idx; ws->nb_bound = a.nb_bound; ws->nb_unit = a.nb_unit;
    ws->rsq = rsq_tab;
}
#endif
