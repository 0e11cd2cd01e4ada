// pw_kernels.hip -- gfx950 kernels and the C ABI of libpywindow_hip.so
// (include/pywindow_amd.h).
//
// One persistent workgroup analyses one (frame, molecule) unit at a time: its
// coordinates, radii and the optimiser state live in LDS for the unit's whole
// lifetime (UnitShared, pw_unit.hpp), units are handed out by an atomic work
// counter so uneven optimiser iteration counts do not idle CUs, and one launch
// covers every unit of a trajectory.  The arithmetic is FP64 vector ALU work with
// exact wave-level (value, index) min reductions; there is nothing GEMM-shaped
// here and MFMA is not used.
//
// Built with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (fused
// multiply-adds only where the source writes pw_fma).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <mutex>
#include <new>

#include "../../include/pywindow_amd.h"
#include "pw_host.hpp"
#include <vector>
// (no PW_TEAM_STATE_IN_LDS here: this file's stage functions serve kernels that keep their team state in LDS -- chains,
// average diameter -- AND kernels that keep it on the stack -- the window search, whose out-of-line stage
// function spills five times as much when its table of pointers is an LDS object: 269 against 55 scratch stores)
#include "pw_unit.hpp"
#include "pw_launch.hpp"

using namespace pw;

// pw_hostpath.cpp (g++): the unit pipeline for a one-lane team, host threads over the units
extern "C" int pw_hostpath_run(const pw_batch_in* in, unsigned stages, pw_unit_out* out, const pw_params* prm, int p_cap,
                               int threads, pw_unit_debug* dbg, pw_extra_window* xw, unsigned xw_cap, unsigned* xw_count);
extern "C" int pw_hostpath_default_threads(void);
// pw_kernels_big.hip: the same source with the team's shared block in global memory (molecules beyond LDS)
extern "C" size_t pw_internal_big_block_bytes(int nmax, int p_cap);
extern "C" int pw_internal_big_launch(void* stream, int grid, long n_units, const long* atom_offset, const double* xyz,
                                      const double* vdw, const double* mass, unsigned stages, int nmax, const PwWsArgs* wsa,
                                      unsigned char* blocks, size_t block_bytes, unsigned long long* counter, pw_unit_out* out,
                                      const pw_params* prm, const unsigned* rsq_tab, int vstride);

namespace {

thread_local char g_err[512] = "";
void set_err(const char* what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(call)                         \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            set_err(#call, e_);               \
            return PW_E_HIP;                  \
        }                                     \
    } while (0)

// first statement of every entry point that touches the device: work on the context's device, give
// the calling thread its own device back on every return path
#define PW_ON_DEVICE(dev)                     \
    DeviceScope dev_scope_;                   \
    HIP_TRY(dev_scope_.enter(dev))

// Successive analyses rotate through up to PW_SETS sets of (result buffer, queue, slots, streams,
// events, team workspaces): that many analyses can be in flight, the optimiser chains of the later
// ones filling the SIMDs that the long tails of the earlier ones leave idle.
constexpr int PW_SETS = 4;      // (2 + 2 x PW_SETS streams, each needs a hardware queue of its own: GPU_MAX_HW_QUEUES = 12)

// static LDS of an analysis kernel (team state: UnitShared, parameters, the unit slot): what every
// launch plan leaves free beside its dynamic request
constexpr size_t PW_KERNEL_STATIC_LDS = 1024;

constexpr unsigned MASK_ANY = 0xffffffffu;
constexpr unsigned MASK_CHAINS = PW_STAGE_BASIC | PW_STAGE_OPT | PW_STAGE_MERGE;
constexpr unsigned MASK_AVERAGE = PW_STAGE_AVG | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;
constexpr unsigned MASK_WINDOWS = PW_STAGE_WINDOWS | PW_STAGE_REUSE_OPT | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;
// ... and the window launch that runs the average-diameter stage as well (round 6: the pipeline's default)
constexpr unsigned MASK_WINDOWS_AVG = MASK_WINDOWS | PW_STAGE_AVG;

// Neighbour tables of the sampling sphere, one block per vector count P (pw_unit.hpp: nb_build_point): the
// P unit vectors go to LDS, every thread tabulates the rows of its points.
__global__ void __launch_bounds__(256) pw_nb_build_kernel(unsigned* __restrict__ off, unsigned short* __restrict__ idx,
                                                          double* __restrict__ bound, double* __restrict__ unit) {
    __shared__ double ux[PW_NB_PMAX], uy[PW_NB_PMAX], uz[PW_NB_PMAX];
    // (large P first: their blocks run longest)
    const int P = PW_NB_PMAX - (int)blockIdx.x;
    Sphere sp;
    sp.init(1.0, P);
    for (int k = threadIdx.x; k < P; k += blockDim.x) sp.point(k, &ux[k], &uy[k], &uz[k]);
    __syncthreads();
    const unsigned first = nb_dense_offset(P);
    if (threadIdx.x == 0) off[P] = first;
    // (the unit vectors themselves: team_sphere_points scales them instead of computing sines and cosines per unit)
    if (unit)
        for (int k = threadIdx.x; k < P; k += blockDim.x) {
            double* q = unit + 3 * ((size_t)first + k);
            q[0] = ux[k]; q[1] = uy[k]; q[2] = uz[k];
        }
    for (int k = threadIdx.x; k < P; k += blockDim.x)
        nb_build_point(P, k, ux, uy, uz, idx + (size_t)(first + k) * PW_NB_K, bound + first + k);
}

// MASK: the stage bits this instantiation can execute (the run-time mask is ANDed with it), so
// the launches of the pipeline carry only the code -- and the registers -- of their own stages
// Register budget: two waves per SIMD (three for the average-diameter launch).  A one-wave-per-SIMD
// build (amdgpu_waves_per_eu(1, 1): 256 VGPRs + AGPRs, no VGPR spills) was tried for the optimiser
// chains and rejected: such a wave owns more than half of its SIMD's register file, so no wave of the
// window launch (256 VGPRs) can share the SIMD with it, and sharing SIMDs between the launches is
// what the pipeline lives on.  The spills that remain are outside the hot loops (0.2 % of the
// chains' instructions, profiles/r02_*).
#ifndef PW_A_PRIO
#define PW_A_PRIO 3
#endif
#ifndef PW_OCC
#define PW_OCC 2
#endif
#ifndef PW_OCC8
#define PW_OCC8 1
#endif
#ifndef PW_OCC_A
#define PW_OCC_A PW_OCC
#endif
// A batch whose coordinates are still arriving (pw_resident_stream_*): `ready` counts the units whose
// coordinates are on the device.  The launches that read coordinates without going through the hand-off queue --
// optimiser chains, average diameter -- take units in index order and wait (bounded) until theirs is there, so the
// analysis can be launched before the reader has decoded the first frame.  ready == nullptr: everything is there.
__device__ inline bool wait_for_unit(const unsigned long long* ready, long u, int* error_flag, long long limit) {
    if (!ready) return true;
    long long t0 = wall_clock64();
    unsigned long long seen = ~0ull;
    for (;;) {
        const unsigned long long have = __hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((long)have > u && !(have >> 63)) break;
        __builtin_amdgcn_s_sleep(16);
        if (have != seen) { seen = have; t0 = wall_clock64(); }     // (the limit is on time WITHOUT an append)
        // the limit (5 s by default: this side waits for the HOST -- a reader, a disk): the host stopped appending;
        // the top bit: the host gave the batch up (pw_resident_free of an incomplete one)
        if ((have >> 63) || wall_clock64() - t0 > limit) {
            if (error_flag) atomicExch(error_flag, (have >> 63) ? 3 : 2);      // (the cause: check_queue_error reports it)
            return false;
        }
    }
    // (system scope: the coordinates were written by the copy engine, not by a wave of this device)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return true;
}
template <int NW, unsigned MASK>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? PW_OCC8 : (MASK == PW_KERNEL_AVERAGE ? 3 : (MASK == MASK_CHAINS ? PW_OCC_A : PW_OCC)))
pw_analyse_kernel(long n_units, const long* __restrict__ atom_offset, const double* __restrict__ xyz,
                  const double* __restrict__ vdw, const double* __restrict__ mass, unsigned stages,
                  int nmax, int nrot, int nlb, int nframes, int lean, PwWsArgs wsa, unsigned long long* counter,
                  pw_unit_out* __restrict__ out, int role, UnitQueue* queue, int* __restrict__ slots,
                  pw_params prm_in, const unsigned* __restrict__ rsq_tab, int vstride,
                  const unsigned long long* __restrict__ ready, const unsigned char* __restrict__ tmpl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ long s_unit;
    // the team's table of pointers and its parameters live in LDS (PW_TEAM_STATE_IN_LDS, pw_unit.hpp): the stage
    // functions are out of line and take them by reference -- on the stack they were 600 bytes of scratch per
    // lane that every look-up went through
    __shared__ UnitShared s_sh;
    __shared__ pw_params s_prm;
    static_assert(sizeof(UnitShared) + sizeof(pw_params) + 16 <= PW_KERNEL_STATIC_LDS, "static LDS of the analysis kernels");
    using T = DeviceTeam<NW>;
    // the optimiser chains are latency-bound and on the critical path: when one shares a SIMD
    // with a bulk wave of another launch it must win the issue arbitration
    if (role == PW_ROLE_PRODUCER) {
        __builtin_amdgcn_s_setprio(PW_A_PRIO);
        if (threadIdx.x == 0) atomicAdd(&queue->started, 1);
    }
    TeamWorkspace* ws = (TeamWorkspace*)wsa.ws + blockIdx.x;
#ifdef PW_PROFILE
    if (threadIdx.x < 32) pw_prof_lds[threadIdx.x] = 0;     // (the team's timers: summed here, flushed when it leaves)
#ifdef PW_BARRIER_PROF
    if (threadIdx.x < 8) pw_bar_acc[threadIdx.x] = 0;
#endif
#endif
    if (threadIdx.x == 0) {
        s_sh.carve(lds, nmax, nrot, nlb, nframes, lean, wsa.p_cap);     // as planned by the host (plan_launch)
        s_prm = prm_in;
        bind_workspace(ws, wsa, blockIdx.x, rsq_tab, team_slab_bytes(wsa.p_cap), team_adj_words(wsa.p_cap));
    }
    __syncthreads();
    // chains and average diameter look their pointers up in LDS; the window search keeps them on its stack (see above)
    constexpr bool STATE_IN_LDS = MASK == MASK_CHAINS || MASK == PW_KERNEL_AVERAGE;
    UnitShared sh_stack;
    if (!STATE_IN_LDS) sh_stack.carve(lds, nmax, nrot, nlb, nframes, lean, wsa.p_cap);
    UnitShared& sh = STATE_IN_LDS ? (UnitShared&)s_sh : sh_stack;
    const pw_params& prm = STATE_IN_LDS ? (const pw_params&)s_prm : prm_in;
    for (;;) {
        if (role == PW_ROLE_CONSUMER) {
            if (threadIdx.x == 0) {
                long pos = (long)atomicAdd(&queue->head, 1ull);
                long u = -1;
                if (pos < n_units) {
                    long long t0 = wall_clock64();
#ifdef PW_PROFILE
                    const long long t_begin = t0;
#endif
                    unsigned long long seen = ~0ull;
                    int spins = 0;
                    for (;;) {
                        int v = __hip_atomic_load(&slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v >= 0) {
                            u = v;
#ifdef PW_PROFILE
                            // (diagnostic builds: what a consumer team spends waiting for its next unit, slot 1)
                            atomicAdd(&pw_prof_lds[1], (unsigned long long)(wall_clock64() - t_begin));
#endif
                            break;
                        }
                        __builtin_amdgcn_s_sleep(32);
                        // (the producer gave up -- its units never arrived: nothing more will be published)
#ifndef PW_NO_CONSUMER_ERROR_CHECK
                        if (__hip_atomic_load(&queue->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
#endif
                        // The limit is on time WITHOUT A PUBLICATION by the optimiser launch (250 ms by default; a chain
                        // publishes every few microseconds, the slowest takes 2 ms): a wait during which the producers
                        // keep coming in -- a device shared with another tenant, a batch of long chains -- is never cut
                        // short, a producer launch that is not running at all is noticed within the limit.
                        if ((++spins & 15) == 0) {
                            const unsigned long long pub = __hip_atomic_load(&queue->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (pub != seen) { seen = pub; t0 = wall_clock64(); }
                        }
                        if (wall_clock64() - t0 > wsa.wait_ticks) {
                            atomicExch(&queue->error, 1);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                s_unit = u;
            }
        } else {
            if (threadIdx.x == 0) {
                long u = (long)atomicAdd(counter, 1ull);
                if (u < n_units && !wait_for_unit(ready, u, &queue->error, wsa.stream_wait_ticks)) u = n_units;
                s_unit = u < n_units ? u : -1;
            }
        }
        __syncthreads();
        long u = s_unit;
        __syncthreads();
        if (u < 0) break;
        long a0 = atom_offset[u];
        int n = (int)(atom_offset[u + 1] - a0);
        // vstride 0: one molecule type, vdw / mass hold a single template (per-trajectory constants)
        const long v0 = a0 * vstride;
        if (threadIdx.x == 0) ws->unit = u;     // (read by the debug capture only; ordered by load_unit's barrier)
        // (tmpl: the radius groups of a one-molecule-type batch, worked out once on the host -- pw_unit.hpp: load_unit)
        analyse_unit<T, MASK>(sh, ws, n, xyz + 3 * a0, vdw + v0, mass + v0, stages & MASK, out + u, prm, vstride == 0 ? tmpl : nullptr);
        if (role == PW_ROLE_PRODUCER) {
            // analyse_unit ended with a team barrier; thread 0 wrote the record
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                long pos = (long)atomicAdd(&queue->tail, 1ull);
                __hip_atomic_store(&slots[pos], (int)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef PW_PROFILE
    __syncthreads();
    if (threadIdx.x < 32 && pw_prof_lds[threadIdx.x]) atomicAdd(&ws->prof[threadIdx.x], pw_prof_lds[threadIdx.x]);
#endif
}

// Gate: one wave, no LDS.  Holds the stream it is launched on until every team of the
// optimiser launch is resident, so that launches queued behind it cannot take the LDS those
// teams need.  It can never block them itself, and its wait is bounded.
// Its limit (PW_WAIT_LIMIT_MS, 250 ms) is on time WITHOUT A TEAM STARTING; what an expiry means is said where it is handled.
__global__ void pw_gate_kernel(UnitQueue* queue, int expected, unsigned long long* timeouts, long long limit) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    int seen = -1;
    for (;;) {
        const int started = __hip_atomic_load(&queue->started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (started >= expected) break;
        __builtin_amdgcn_s_sleep(16);
        if (started != seen) { seen = started; t0 = wall_clock64(); }      // (the limit is on time without a team starting)
        if (wall_clock64() - t0 > limit) {
            // Counted either way.  Teams HAVE started and then none for a whole limit: the device is busy with something
            // else (another tenant, earlier analyses' chains); the launches behind the gate may go -- the window teams are
            // capped at five per four CUs (launch_pipeline), so the optimiser teams that are still to come always find
            // wave slots and LDS, and the consumers' own limit watches the publications from here on.  NOT ONE team has
            // started: the optimiser launch is not running at all -- its stream is behind something that does not move --
            // and consumers would only wait for it: the analysis is given up (cause 4, PW_E_TIMEOUT).
            atomicAdd(timeouts, 1ull << 32);
            if (started == 0) atomicCAS(&queue->error, 0, 4);
            break;
        }
    }
}

// Tail gate: holds the optimiser launch of analysis k+1 until all but the slowest few chains
// of analysis k have been published, so that the two launches overlap only where the older one
// leaves most SIMDs idle.  One wave, bounded wait.
__global__ void pw_tail_gate_kernel(const UnitQueue* prev, unsigned long long need, unsigned long long* timeouts) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    while (__hip_atomic_load(&prev->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(32);
        // 20 ms: only an optimisation, never a dependency (counted: pw_context_gate_timeouts)
        if (wall_clock64() - t0 > 2000000ll) { atomicAdd(timeouts, 1ull); break; }
    }
}

// The same for the window launch: analysis k+1's consumers start when all but the last few
// units of analysis k have been TAKEN by a team (its queue head has advanced that far).
__global__ void pw_head_gate_kernel(const UnitQueue* prev, unsigned long long need, unsigned long long* timeouts) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    while (__hip_atomic_load(&prev->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > 2000000ll) { atomicAdd(timeouts, 1ull << 16); break; }     // 20 ms, as above
    }
}

// Probe of pw_context_create: K of these, one per stream of the pipeline, must be able to RUN AT THE SAME
// TIME -- every one waits (bounded) until all K have started.  Streams that share a hardware queue
// (GPU_MAX_HW_QUEUES too small, or exported after the process initialised HIP) run one after the other and
// the probe says so; the pipeline's gate kernels would then sit in front of the launches they wait for.
__global__ void pw_probe_kernel(int* started, int expected, int* together) {
    if (threadIdx.x != 0) return;
    atomicAdd(started, 1);
    long long t0 = wall_clock64();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 1000000ll) return;     // 10 ms
    }
    atomicAdd(together, 1);
}

// Fine-grained entry: min_i(|r_i - p| - vdw_i) and its first argmin for arbitrary
// points p (reference pore_diameter(elements, coordinates, com=p)/2,
// utilities.py:375-388).  One lane per point, atoms streamed from global memory.
// DBSCAN(eps, min_samples = 5) of one point set by one team (pw_dbscan): the routine of the window search on
// its own.  pts: three arrays of n (x | y | z).  lds_bytes > 0: adjacency rows, compacted points and labels
// in that much dynamic LDS when they fit (what the pipeline does for CC3); 0: everything in global memory.
template <int NW>
__global__ void __launch_bounds__(NW * 64) pw_dbscan_kernel(const double* __restrict__ pts, int n, double eps,
                                                            int lds_bytes, TeamWorkspace* ws, int p_cap,
                                                            unsigned long long* adj_base, const int* __restrict__ ident,
                                                            int* __restrict__ labels, int* __restrict__ n_clusters) {
    using T = DeviceTeam<NW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ unsigned long long bits[3][PW_DBSCAN_MAX / 64];
    if (threadIdx.x == 0) { ws->adj = adj_base; ws->p_cap = p_cap; }
    __syncthreads();
    ScratchArena arena;
    arena.cur = (unsigned char*)lds;
    arena.left = (size_t)lds_bytes;
    int* lab = (int*)arena.take((size_t)n * 4);
    if (!lab) lab = labels;
    auto PT = [n](int k, int c) { return c * n + k; };
    int k = team_dbscan<T>((PW_LDS unsigned long long*)bits[0], (PW_LDS unsigned long long*)bits[1],
                           (PW_LDS unsigned long long*)bits[2], arena, ws, pts, PT, ident, n, eps, lab);
    if (lab != labels)
        for (int i = threadIdx.x; i < n; i += NW * 64) labels[i] = lab[i];
    if (threadIdx.x == 0) *n_clusters = k;
}

// numpy's float64 add.reduce of one array by one team (pw_pairwise_sum): the team-shared scratch of the
// sum either in LDS or -- mode bit 1 -- in global memory (the fallback of molecules too large for LDS)
template <int NW>
__global__ void __launch_bounds__(NW * 64) pw_pairwise_sum_kernel(const double* __restrict__ a, long n,
                                                                   double* __restrict__ scratch, int use_global,
                                                                   double* __restrict__ out) {
    using T = DeviceTeam<NW>;
    __shared__ int s_tab[324];
    __shared__ double s_acc[8 * 160];
    __shared__ double s_leaf[256];
    __shared__ double s_slot;
    int* tab = use_global ? (int*)(scratch + 8 * 160 + 256) : (int*)s_tab;
    double* acc = use_global ? scratch : (double*)s_acc;
    double* leaf = use_global ? scratch + 8 * 160 : (double*)s_leaf;
    double r = np_sum_team<T>(a, (int)n, tab, acc, leaf, (double*)&s_slot);
    if (threadIdx.x == 0) *out = r;
}

__global__ void pw_point_gap_kernel(long n_points, const long* __restrict__ unit_of_point,
                                    const double* __restrict__ points,
                                    const long* __restrict__ atom_offset,
                                    const double* __restrict__ xyz, const double* __restrict__ vdw,
                                    double* __restrict__ gap, int* __restrict__ arg, int vstride) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_points) return;
    long u = unit_of_point[q];
    long a0 = atom_offset[u];
    int n = (int)(atom_offset[u + 1] - a0);
    double px = points[3 * q], py = points[3 * q + 1], pz = points[3 * q + 2];
    double pp = sq3(px, py, pz);
    double best = PW_INF;
    int bi = 0;
    for (int i = 0; i < n; ++i) {
        double x = xyz[3 * (a0 + i)], y = xyz[3 * (a0 + i) + 1], z = xyz[3 * (a0 + i) + 2];
        double xx = sq3(x, y, z);
        double g = pw_fma(z, pz, pw_fma(x, px, y * py));
        double d2 = pw_m2add(g, xx) + pp;
        double d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
        double v = d - vdw[a0 * vstride + i];
        if (v < best) { best = v; bi = i; }
    }
    gap[q] = best;
    arg[q] = bi;
}

// pw_div_r against the division it replaces (pw_common.hpp), on operand pairs made up on the device from a counter:
// mode 0 magnitudes as the optimisers see them (2^-40 .. 2^40), 1 any bit pattern (denormals, infinities, NaNs, zeros:
// the guarded path), 2 divisors whose significand is all ones or nearly (the hard case of reciprocal-based division),
// 3 quotients that are exactly representable (a = q * b with short q, b).  out[0] = pairs that differ, out[1..3] =
// the bits of the first such pair and of the wrong quotient.
__global__ void pw_div_check_kernel(unsigned long long n, int mode, unsigned long long seed, unsigned long long* out) {
    const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    auto mix = [](unsigned long long z) {
        z += 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto as_d = [](unsigned long long u) { union { unsigned long long u; double d; } c; c.u = u; return c.d; };
    auto as_u = [](double d) { union { unsigned long long u; double d; } c; c.d = d; return c.u; };
    for (unsigned long long i = i0; i < n; i += stride) {
        const unsigned long long r1 = mix(seed + 2 * i), r2 = mix(seed + 2 * i + 1), r3 = mix(r1 ^ r2);
        double a, b;
        if (mode == 1) {
            a = as_d(r1);
            b = as_d(r2);
            if ((r3 & 15) == 0) a = as_d(r1 & 0x800fffffffffffffull);                 // denormal / zero dividend
            if ((r3 & 0xf0) == 0) b = as_d(r2 & 0x800fffffffffffffull);
            if ((r3 & 0xf00) == 0) a = as_d(r1 & 0x8000000000000000ull);             // +-0
        } else {
            const unsigned long long ea = 1023 - 40 + (r3 % 81), eb = 1023 - 40 + ((r3 >> 20) % 81);
            a = as_d((r1 & 0x800fffffffffffffull) | (ea << 52));
            b = as_d((r2 & 0x800fffffffffffffull) | (eb << 52));
            if (mode == 2) b = as_d(as_u(b) | (0x000fffffffffffffull & ~((r3 >> 40) & 0xff)));
            if (mode == 3) {
                const double q = as_d((r1 & 0x800ffffff8000000ull) | (ea << 52));
                b = as_d((r2 & 0x800ffffff0000000ull) | (eb << 52));
                a = q * b;                                                             // exact: 26 x 24 bits
            }
        }
        const double want = a / b;
        const double got = pw_div_r(a, b, pw_recip_hw(b));
        const bool same = as_u(want) == as_u(got) || (want != want && got != got);
        if (!same) {
            if (atomicAdd(&out[0], 1ull) == 0ull) { out[1] = as_u(a); out[2] = as_u(b); out[3] = as_u(got); }
        }
    }
}

// Start of a pipeline launch: the record buffer, the hand-off queue with its slots and the three work
// counters of the launch's set, all in ONE small kernel (six hipMemsetAsync calls took 50 us of
// stream time each -- a tenth of the step of a small batch).
__global__ void pw_reset_kernel(unsigned long long* __restrict__ out8, long n8, UnitQueue* queue, int* __restrict__ slots,
                                long n_units, unsigned long long* ca, unsigned long long* cb,
                                unsigned long long* cc, unsigned long long* cd, unsigned* xw_count) {
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = i0; i < n8; i += stride) out8[i] = 0ull;
    for (long i = i0; i < n_units; i += stride) slots[i] = -1;
    if (i0 == 0) {
        queue->tail = 0; queue->head = 0; queue->error = 0; queue->started = 0;
        *ca = 0; *cb = 0; *cc = 0; *cd = 0;
        *xw_count = 0u;
    }
}

// pw_analysis_debug: point every team workspace at the capture buffer (or away from it)
__global__ void pw_set_debug_kernel(TeamWorkspace* ws, int blocks, pw_unit_debug* dbg) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < blocks) ws[b].dbg_base = dbg;
}

}  // namespace

static std::atomic<unsigned long long> g_retries_total{0};
struct pw_context {
    unsigned long long retries = 0;   // analyses repeated after PW_E_TIMEOUT (pw_context_retries)
    long long wait_ticks;             // PW_WAIT_LIMIT_MS (default 250): a launch waits this long for another one WITHOUT PROGRESS
    long long stream_wait_ticks;      // PW_STREAM_LIMIT_MS (default 5000): ... and for the host's next append to a streamed batch
    int timeout_repeats;              // PW_TIMEOUT_REPEATS (default 2): how often pw_analysis_batch repeats an analysis after PW_E_TIMEOUT
    int device;
    hipStream_t stream;      // main stream (launch order, timing events)
    hipStream_t aux;         // second stream: stages that do not depend on the optimiser
    int n_cu;
    size_t lds_per_cu;
    TeamWorkspace* ws;
    int ws_blocks;
    unsigned char* slab;     // ws_blocks x team_slab_bytes(p_cap)
    unsigned long long* adj;
    int adj_blocks;
    int p_cap;               // sampling vectors per molecule the slabs / adjacency rows are sized for (0: none yet)
    int p_cap_min;           // ... and at least this many (raised when a unit asked for more than the knobs imply)
    std::vector<pw_extra_window>* extra;   // windows beyond PW_W_MAX of the records fetched last (host copy)
    unsigned long long* counter;   // 4 work counters
    UnitQueue* queue;
    int* slots;
    long slots_cap;
    hipStream_t prod;        // optimiser launch of the overlapped pipeline (this launch's of prods[])
    hipStream_t prods[PW_SETS];
    // successive pipeline launches alternate between two sets of (result buffer, queue,
    // slots, events), so the optimiser chains of launch k+1 run beside the window tail of k
    hipEvent_t ev_reset[PW_SETS], ev_prod[PW_SETS], ev_gate[PW_SETS], ev_join[PW_SETS], ev_done[PW_SETS];
    int done_valid[PW_SETS];
    hipStream_t cons[PW_SETS];   // gate + window launch of the pipeline, one stream per buffer set
    hipEvent_t ev_head[PW_SETS]; // head gate of the launch using set b has run
    int head_valid[PW_SETS];
    int nsets;               // analyses in flight at most (PW_SETS_IN_FLIGHT, 2..PW_SETS), 0: chosen per batch size
    int cur_sets;            // ... of the current layout of the team workspaces
    int head_pct;            // PW_HEAD_GATE (default 85; 0 = window launches strictly one after another)
    hipEvent_t ev_tail[PW_SETS];   // tail gate of the launch using set b has run
    int tail_valid[PW_SETS];
    long last_units[PW_SETS];
    const void* last_res[PW_SETS];   // batch of the latest launch on each set (identity only, never dereferenced)
    int tail_pct;            // PW_TAIL_GATE: start the next optimiser launch at this % published
                             // (default 80; 0 = strictly one optimiser launch at a time)
    int flip;                // buffer set of the latest pipeline launch (-1: none yet)
    int need_fork;           // main stream carries work the next pipeline launch must wait for
    UnitQueue* cur_queue;
    int* cur_slots;
    hipEvent_t ev0, ev1, ev_fork;
    int fused;               // PW_FUSED=1: one launch per analysis instead of the pipeline (also chosen when the
                             // streams of the pipeline do not run concurrently, see pw_context_create)
    int host_threads;        // device == -1 (the explicit host path, pw_hostpath.cpp): threads over the units
    int concurrent_streams;  // how many of the pipeline's 2 + 2 x PW_SETS streams ran at the same time in the probe
    int c_waves;             // waves per team in the window launch (PW_C_WAVES, default 4)
    pw_params prm;           // knobs of find_windows / find_average_diameter
    unsigned* rsq_tab;       // VRSQRT14PD table on the device (numpy's arccos, pw_math.hpp)
    unsigned* nb_off;        // neighbour tables of the sampling sphere for P = PW_NB_PMIN .. PW_NB_PMAX (pw_unit.hpp)
    unsigned short* nb_idx;
    double* nb_bound;
    double* nb_unit;         // the unit vectors of every tabulated P (50 MB)
    // team workspaces of the pipeline: C0 | A0 | B | A1 | C1, each region sized for the largest grid
    // any launch on this context has asked for so far.  The layout only changes when a maximum
    // grows, and growing synchronises the device first, so two launches in flight -- which may have
    // different plans (other batch size, other stage mask) -- never share a team workspace.
    int max_a, max_b, max_c;
    hipEvent_t ev_ext;       // ordering against a caller's stream (pw_resident_results_ready)
    hipEvent_t ev_t[3][2];   // pw_resident_stage_times: start / stop of the chains, average and window launches
    int timing;              // record them during the next pipeline launch
    int timed_avg;           // ... and whether that launch had an average-diameter launch of its own (else the stage ran in the window teams)
    pw_unit_debug* dbg;      // per-unit stage capture of the current debug analysis, else null
    hipStream_t rb_stream;   // the periodic re-assembly's own stream, highest priority (pw_internal_rebuild_stream), created on first use
    void* pool;              // device scratch kept between calls (the team slabs of the periodic re-assembly)
    size_t pool_bytes;
    // device blocks of freed batches, kept for the next upload (hipMalloc / hipFree cost tens of microseconds
    // each and hipFree waits for the device: a trajectory that goes through in pieces would pay both per piece)
    struct Block { void* p; size_t bytes; };
    std::vector<Block>* blocks;
    size_t blocks_bytes;
    unsigned char* bigmem;   // team blocks of the global-memory analysis (pw_kernels_big.hip), grown on demand
    size_t bigmem_bytes;
    void* pinned;            // page-locked host staging buffer handed to the reader (pw_context_pinned)
    size_t pinned_bytes;
    unsigned long long* ready_vals;   // page-locked, device-mapped ring of `ready` counters: a streamed batch takes one, the host
    unsigned long long* ready_dev;    // raises it after each copy, the waiting teams poll it (pw_resident_stream_*); ready_dev:
    unsigned ready_at;                // the ring's device address
    std::recursive_mutex* mu;      // held by every entry point for the duration of the call (pw_host.hpp)
};

// test hook (tests/test_gpu_api.py): out4 = {pairs that differ, a, b, wrong quotient of the first}; see the kernel
extern "C" int pw_internal_div_check(int device, unsigned long long n, int mode, unsigned long long seed, unsigned long long* out4) {
    if (!out4 || device < 0) return PW_E_BAD_ARG;
    int old = 0;
    if (hipGetDevice(&old) != hipSuccess || hipSetDevice(device) != hipSuccess) return PW_E_HIP;
    unsigned long long* d = nullptr;
    hipError_t e = hipMalloc(&d, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pw_div_check_kernel, dim3(2048), dim3(256), 0, 0, n, mode, seed, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out4, d, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    (void)hipSetDevice(old);
    return e == hipSuccess ? PW_OK : PW_E_HIP;
}

extern "C" void pw_internal_lock(pw_context* c) { if (c && c->mu) c->mu->lock(); }
extern "C" void pw_internal_unlock(pw_context* c) { if (c && c->mu) c->mu->unlock(); }

// a batch "resident" on the host (contexts created with device = -1)
struct HostBatch {
    std::vector<int64_t> offset;
    std::vector<double> xyz, vdw, mass;
    std::vector<pw_unit_out> out;
    int64_t template_atoms;
};

struct pw_resident {
    HostBatch* host;         // non-null: the batch of a host context, everything below unused
    long n_units;
    long n_atoms;
    int nmax;
    int vstride;             // 1: d_vdw / d_mass per atom; 0: one template of nmax atoms for every unit
    unsigned long long* d_ready;   // streamed batch (pw_resident_stream_begin): units whose coordinates have arrived -- a counter
    unsigned long long* h_ready;   // in page-locked host memory that the device reads (d_ready: its device address); else null
    long ready_units;              // ... as the host has raised it
    int pending;                   // a launch asked for while units were missing, in a shape that shares the stream the
    unsigned pending_stages;       // appends copy on (one launch per analysis): made by the append that completes the batch
    long* d_offset;
    double* d_xyz;
    double* d_vdw;
    double* d_mass;
    unsigned char* d_tmpl;   // one molecule type: the radius groups of the template (pw_unit.hpp: template_groups_build), else null
    void* block;             // one device block holds every array of an uploaded batch (from the context's cache)
    size_t block_bytes;
    void* parts[5];          // ... or, for a batch assembled on the device (pw_resident_from_cells), one block per array
    size_t part_bytes[5];
    pw_unit_out* d_out;      // result records of the latest launch (= d_outs[cur])
    pw_unit_out* d_outs[PW_SETS];
    pw_extra_window* d_xw[PW_SETS];   // windows beyond PW_W_MAX written by the launch into d_outs[k] ...
    unsigned* d_xw_count;             // ... and how many (PW_SETS counters)
    unsigned xw_cap;
    int nbuf;                // result buffers in rotation (= the context's sets in flight at upload)
    int cur;
    hipEvent_t ev_read[PW_SETS];   // a caller's stream has finished reading d_outs[k] (pw_resident_results_release)
    int read_valid[PW_SETS];
    int written_set[PW_SETS];      // pipeline set of the launch that last wrote d_outs[k], -1: none / not a pipeline launch
};

static int block_take(pw_context* c, size_t bytes, void** out, size_t* got) {
    bytes = (bytes + 65535) & ~(size_t)65535;
    int best = -1;
    for (int i = 0; i < (int)c->blocks->size(); ++i) {
        const size_t b = (*c->blocks)[i].bytes;
        if (b >= bytes && b <= 2 * bytes + (1u << 20) && (best < 0 || b < (*c->blocks)[best].bytes)) best = i;
    }
    if (best >= 0) {
        *out = (*c->blocks)[best].p;
        *got = (*c->blocks)[best].bytes;
        c->blocks_bytes -= *got;
        c->blocks->erase(c->blocks->begin() + best);
        return PW_OK;
    }
    HIP_TRY(hipMalloc(out, bytes));
    *got = bytes;
    return PW_OK;
}
static void block_give(pw_context* c, void* p, size_t bytes) {
    if (!p) return;
    if (c && c->blocks && c->blocks->size() < 96 && c->blocks_bytes + bytes <= ((size_t)8 << 30)) {
        c->blocks->push_back({p, bytes});
        c->blocks_bytes += bytes;
    } else {
        (void)hipFree(p);
    }
}

// the same cache for the other translation units (periodic re-assembly: its seventeen temporaries per call)
extern "C" int pw_internal_block_take(pw_context* c, size_t bytes, void** out, size_t* got) {
    if (!c || !out || !got || c->device < 0) return PW_E_BAD_ARG;
    return block_take(c, bytes ? bytes : 8, out, got);
}
extern "C" void pw_internal_block_give(pw_context* c, void* p, size_t bytes) { block_give(c, p, bytes); }

// sampling-vector capacity the next launch needs: what the adjust knobs imply (pw_unit.hpp: params_p_cap)
// or what a unit of an earlier analysis asked for, whichever is larger
// analyses in flight (= buffer sets of a batch) when PW_SETS_IN_FLIGHT does not say: see launch_pipeline
static int auto_sets(long n_units) { return n_units <= 600 ? 4 : (n_units <= 3000 ? 3 : 2); }
// the share of the previous launch's units that must have been published / taken before the next optimiser / window
// launch of the same batch starts, where the environment does not say: round 5's re-sweep on the faster chains --
// 1000 units 1.26 ms at 70 % against 1.29 at 50 %, 250 units 0.62-0.66 ms at 50 % against 0.71 at 70 %
// (profiles/r05_resweep.txt)
static int gate_pct(int configured, long n_units) { return configured >= 0 ? configured : (n_units <= 600 ? 50 : 70); }

static int wanted_p_cap(const pw_context* c) {
    int p = params_p_cap(c->prm.adjust_windows, c->prm.adjust_average);
    if (c->p_cap_min > p) p = round_p_cap(c->p_cap_min);
    return p;
}
// capacities beyond this are "large": such launches run as single launches on a limited number of teams
// (the slabs and adjacency rows of thousands of pipeline teams would not fit), and the workspace is rebuilt
// small again when the knobs go back
constexpr int PW_P_CAP_PIPELINE = 8448;          // round_p_cap(2100 x 4): up to adjust = 4 the pipeline runs
static size_t team_ws_bytes(int p_cap, bool with_adj) {
    return sizeof(TeamWorkspace) + team_slab_bytes(p_cap) + (with_adj ? team_adj_words(p_cap) * 8 : 0);
}
static int ensure_workspace(pw_context* c, int blocks, int adj_blocks) {
    const int want = wanted_p_cap(c);
    if (want > c->p_cap || (want < c->p_cap && c->p_cap > PW_P_CAP_PIPELINE)) {
        // the capacity follows the knobs upwards always, downwards only from a large one; everything sized by
        // it is rebuilt for THIS launch's teams (nothing is in flight afterwards)
        HIP_TRY(hipDeviceSynchronize());
        if (c->ws) HIP_TRY(hipFree(c->ws));
        c->ws = nullptr;
        if (c->slab) HIP_TRY(hipFree(c->slab));
        c->slab = nullptr;
        if (c->adj) HIP_TRY(hipFree(c->adj));
        c->adj = nullptr;
        c->ws_blocks = 0;
        c->adj_blocks = 0;
        c->p_cap = want;
    }
    if (c->ws_blocks < blocks) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->ws) HIP_TRY(hipFree(c->ws));
        c->ws = nullptr;
        if (c->slab) HIP_TRY(hipFree(c->slab));
        c->slab = nullptr;
        c->ws_blocks = 0;
        HIP_TRY(hipMalloc((void**)&c->ws, (size_t)blocks * sizeof(TeamWorkspace)));
        HIP_TRY(hipMemset(c->ws, 0, (size_t)blocks * sizeof(TeamWorkspace)));
        HIP_TRY(hipMalloc((void**)&c->slab, (size_t)blocks * team_slab_bytes(c->p_cap)));
        c->ws_blocks = blocks;
    }
    if (c->adj_blocks < adj_blocks) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->adj) HIP_TRY(hipFree(c->adj));
        c->adj = nullptr;
        c->adj_blocks = 0;
        HIP_TRY(hipMalloc((void**)&c->adj, (size_t)adj_blocks * team_adj_words(c->p_cap) * sizeof(unsigned long long)));
        c->adj_blocks = adj_blocks;
    }
    return PW_OK;
}

struct LaunchPlan {
    int nw, nrot, nlb, grid, nframes, lean;
    size_t lds;
};

// team width, LDS carve and grid for one launch.  want_nw: preferred waves per team;
// rot/lb: whether window frames / optimiser states are needed (per wave).
// want_nw: 1 (one wave per unit: stages without bulk loops) or 4.  A molecule whose rotated window frames and optimiser
// blocks do not fit beside each other for four waves gets fewer fit SLOTS (4 -> 2 -> 1: the windows of a unit are then
// fitted in rounds, UnitShared::nslots), not fewer waves -- the bulk stages keep the whole team, and the library carries
// one general kernel shape instead of four.
static int plan_launch(pw_context* c, long n_units, int nmax, int want_nw, bool rot, int lb_per_team,
                       LaunchPlan* p, int nframes = 2, int lean = 0) {
    const size_t max_lds = 160 * 1024 - 256 - PW_KERNEL_STATIC_LDS;
    const int nw = want_nw >= 4 ? 4 : 1;
    for (int nslot = nw < 4 ? nw : 4;; nslot >>= 1) {
        int nrot = rot ? nslot : 0;
        int nlb = lb_per_team < 0 ? nslot : lb_per_team;
        size_t lds = UnitShared::bytes(nmax, nrot, nlb, nframes, lean, wanted_p_cap(c) > c->p_cap ? wanted_p_cap(c) : c->p_cap) + 64;
        if (lds <= max_lds || nslot == 1) {
            if (lds > max_lds) {
                snprintf(g_err, sizeof(g_err), "molecule with %d atoms does not fit in LDS", nmax);
                return PW_E_TOO_LARGE;
            }
            p->nw = nw; p->nrot = nrot; p->nlb = nlb; p->lds = lds; p->nframes = nframes; p->lean = lean;
            break;
        }
    }
    int per_cu = (int)(c->lds_per_cu / (p->lds + PW_KERNEL_STATIC_LDS));
    int wave_cap = 16 / p->nw;   // kernels are built for 2 waves per SIMD
    if (per_cu > wave_cap) per_cu = wave_cap;
    if (per_cu < 1) per_cu = 1;
    long grid = (long)c->n_cu * per_cu;
    if (grid > n_units) grid = n_units;
    if (grid < 1) grid = 1;
    p->grid = (int)grid;
    return PW_OK;
}

// PW_TEMPLATE_GROUPS=0: every unit works its radius groups out itself, as until round 5 (A/B comparisons)
static bool template_groups_on() {
    static const bool on = !(getenv("PW_TEMPLATE_GROUPS") && getenv("PW_TEMPLATE_GROUPS")[0] == '0');
    return on;
}
template <int NW, unsigned MASK>
static int launch_nw(pw_context* c, pw_resident* r, unsigned stages, const LaunchPlan& p, hipStream_t st,
                     int ws_first, int adj_first, int counter_slot, int role, bool reset_counter) {
    auto kern = pw_analyse_kernel<NW, MASK>;
    if (getenv("PW_PLAN_DEBUG"))
        fprintf(stderr, "launch NW=%d mask %x grid %d lds %zu nmax %d units %ld atoms %ld\n", NW, MASK, p.grid, p.lds,
                r->nmax, r->n_units, r->n_atoms);
    // the limit, not a request: set once per kernel and device to the most a launch may ask for, so
    // that concurrent launches from several host threads never lower it under each other
    {
        static std::atomic<unsigned long long> done{0};
        unsigned long long bit = 1ull << (c->device & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024 - 256 - (int)PW_KERNEL_STATIC_LDS));
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    if (reset_counter) HIP_TRY(hipMemsetAsync(c->counter + counter_slot, 0, sizeof(unsigned long long), st));
    PwWsArgs wsa;
    wsa.ws = c->ws + ws_first;
    wsa.slab = c->slab + (size_t)ws_first * team_slab_bytes(c->p_cap);
    wsa.adj = adj_first >= 0 ? c->adj + (size_t)adj_first * team_adj_words(c->p_cap) : (unsigned long long*)nullptr;
    wsa.xwin = r->d_xw[r->cur];
    wsa.xwin_count = r->d_xw_count + r->cur;
    wsa.xwin_cap = r->xw_cap;
    wsa.p_cap = c->p_cap;
    wsa.nb_off = c->nb_off; wsa.nb_idx = c->nb_idx; wsa.nb_bound = c->nb_bound; wsa.nb_unit = c->nb_unit;
    // (a chain's length grows with the molecule: the limit with it -- CC3's 168 atoms: as configured)
    wsa.wait_ticks = c->wait_ticks * (1 + r->nmax / 512); wsa.stream_wait_ticks = c->stream_wait_ticks;
    hipLaunchKernelGGL(kern, dim3(p.grid), dim3(NW * 64), p.lds, st, r->n_units, r->d_offset,
                       r->d_xyz, r->d_vdw, r->d_mass, stages, r->nmax, p.nrot, p.nlb, p.nframes, p.lean, wsa,
                       c->counter + counter_slot, r->d_out, role, c->cur_queue, c->cur_slots, c->prm, c->rsq_tab,
                       r->vstride, (const unsigned long long*)r->d_ready, (const unsigned char*)(template_groups_on() ? r->d_tmpl : nullptr));
    HIP_TRY(hipGetLastError());
    return PW_OK;
}
static int launch_plan(pw_context* c, pw_resident* r, unsigned stages, const LaunchPlan& p, hipStream_t st,
                       int ws_first, int adj_first, int counter_slot, int role = PW_ROLE_PLAIN,
                       bool reset_counter = true) {
    // the three launches of the pipeline have kernels of their own
    if (stages == MASK_CHAINS && p.nw == 1)
        return launch_nw<1, MASK_CHAINS>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
    if (stages == MASK_AVERAGE && p.nw == 4)
        return launch_nw<4, MASK_AVERAGE>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
    if (stages == MASK_WINDOWS && p.nw == 4)
        return launch_nw<4, MASK_WINDOWS>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
    if (stages == MASK_WINDOWS_AVG && p.nw == 4)
        return launch_nw<4, MASK_WINDOWS_AVG>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
    // single launches: one wave per unit for the stages the chains kernel holds (basic, optimised pore), the general
    // four-wave kernel for everything else
    if (p.nw == 1 && (stages & ~MASK_CHAINS) == 0)
        return launch_nw<1, MASK_CHAINS>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
    if (p.nw != 4) {
        snprintf(g_err, sizeof(g_err), "internal: no kernel for %d-wave teams with stages 0x%x", p.nw, stages);
        return PW_E_BAD_ARG;
    }
    return launch_nw<4, MASK_ANY>(c, r, stages, p, st, ws_first, adj_first, counter_slot, role, reset_counter);
}

// the API stream (uploads, downloads, single-launch analyses, timing marks) follows every
// pipeline launch issued so far
static int join_pipeline(pw_context* c) {
    for (int b = 0; b < PW_SETS; ++b)
        if (c->done_valid[b]) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done[b], 0));
    return PW_OK;
}

// ---- device = -1: the explicit host path ------------------------------------------------------------
// the same two retries as launch_and_download: a unit that wants more sampling vectors than the knobs
// imply, more windows than the list holds
static int host_analyse(pw_context* c, const pw_batch_in* in, uint32_t stages, pw_unit_out* out, pw_unit_debug* dbg) {
    if (in->n_units == 0) return PW_OK;
    for (int64_t u = 0; u < in->n_units; ++u) {
        const int64_t n = in->atom_offset[u + 1] - in->atom_offset[u];
        if (n <= 0 || (in->template_atoms > 0 && n != in->template_atoms)) {
            snprintf(g_err, sizeof(g_err), "unit %ld has %ld atoms", (long)u, (long)n);
            return PW_E_BAD_ARG;
        }
    }
    stages &= PW_STAGE_ALL;
    int p_cap = wanted_p_cap(c);
    std::vector<pw_extra_window> xw(1024);
    c->extra->clear();
    for (int attempt = 0; attempt < 4; ++attempt) {
        unsigned count = 0;
        int rc = pw_hostpath_run(in, stages, out, &c->prm, p_cap, c->host_threads, dbg, xw.data(), (unsigned)xw.size(), &count);
        if (rc != PW_OK) { snprintf(g_err, sizeof(g_err), "host path: out of memory"); return rc; }
        long want = 0;
        for (int64_t u = 0; u < in->n_units; ++u)
            if (out[u].status & PW_ST_POINTS_OVERFLOW) {
                if (out[u].n_points > want) want = out[u].n_points;
                if (out[u].n_points_avg > want) want = out[u].n_points_avg;
            }
        if (want > p_cap) { p_cap = round_p_cap(want); c->p_cap_min = p_cap; continue; }
        if (count > xw.size()) { xw.resize(count + count / 2); continue; }
        xw.resize(count);
        std::sort(xw.begin(), xw.end(), [](const pw_extra_window& a, const pw_extra_window& b) {
            return a.unit != b.unit ? a.unit < b.unit : a.index < b.index;
        });
        *c->extra = xw;
        return PW_OK;
    }
    snprintf(g_err, sizeof(g_err), "host path: capacities did not settle");
    return PW_E_TOO_LARGE;
}
#define PW_HOST_UNSUPPORTED(c, what)                                                                       \
    if ((c) && (c)->device < 0) {                                                                          \
        snprintf(g_err, sizeof(g_err), what ": not part of the host path (device = -1 runs the analysis only)"); \
        return PW_E_NO_DEVICE;                                                                             \
    }

extern "C" {

const char* pw_version(void) { return "pywindow_amd 0.1 (gfx950)"; }
const char* pw_last_error(void) { return g_err; }

int pw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pw_context_create(int device, pw_context** out) {
    if (!out) return PW_E_BAD_ARG;
    *out = nullptr;
    if (device == -1) {
        // the explicit host path: no HIP call is made on behalf of this context, ever
        pw_context* h = new (std::nothrow) pw_context();
        if (!h) return PW_E_NOMEM;
        memset(h, 0, sizeof(*h));
        h->device = -1;
        h->prm = default_params();
        h->extra = new (std::nothrow) std::vector<pw_extra_window>();
        h->mu = new (std::nothrow) std::recursive_mutex();
        if (!h->extra || !h->mu) { delete h->extra; delete h->mu; delete h; return PW_E_NOMEM; }
        h->host_threads = pw_hostpath_default_threads();
        h->fused = 1;
        *out = h;
        return PW_OK;
    }
    int n = pw_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        snprintf(g_err, sizeof(g_err), "no usable HIP device (count=%d, requested %d)", n, device);
        return PW_E_NO_DEVICE;
    }
    // PW_CONTEXT_TIMING=1: where the time of this call goes, one line on stderr (the cold start of a process is mostly
    // this call: DESIGN.md section 6)
    const bool ctx_timing = getenv("PW_CONTEXT_TIMING") && getenv("PW_CONTEXT_TIMING")[0] == '1';
    timespec ct0;
    clock_gettime(CLOCK_MONOTONIC, &ct0);
    double ct_last = 0.0;
    char ct_text[400] = "";
    auto ct_mark = [&](const char* what) {
        if (!ctx_timing) return;
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        const double ms = (t.tv_sec - ct0.tv_sec) * 1e3 + (t.tv_nsec - ct0.tv_nsec) * 1e-6;
        const size_t at = strlen(ct_text);
        snprintf(ct_text + at, sizeof(ct_text) - at, " %s %.1f", what, ms - ct_last);
        ct_last = ms;
    };
    PW_ON_DEVICE(device);
    ct_mark("device");
    pw_context* c = new (std::nothrow) pw_context();
    if (!c) return PW_E_NOMEM;
    memset(c, 0, sizeof(*c));
    c->device = device;
    // any failure below: release what exists so far (pw_context_destroy takes a half-built object)
#define CTX_TRY(call)                         \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            set_err(#call, e_);               \
            pw_context_destroy(c);            \
            return PW_E_HIP;                  \
        }                                     \
    } while (0)
    hipDeviceProp_t prop;
    CTX_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    c->lds_per_cu = 160 * 1024;
    ct_mark("properties");
    CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTX_TRY(hipMalloc((void**)&c->counter, (4 * PW_SETS + 3) * sizeof(unsigned long long)));      // per set: chains | windows | average | basic; + 2 for single launches; + the gates' time-outs
    CTX_TRY(hipMemset(c->counter, 0, (4 * PW_SETS + 3) * sizeof(unsigned long long)));
    CTX_TRY(hipMalloc((void**)&c->queue, PW_SETS * sizeof(UnitQueue)));
    CTX_TRY(hipMemset(c->queue, 0, PW_SETS * sizeof(UnitQueue)));
    c->flip = -1;
    c->extra = new (std::nothrow) std::vector<pw_extra_window>();
    c->blocks = new (std::nothrow) std::vector<pw_context::Block>();
    c->mu = new (std::nothrow) std::recursive_mutex();
    if (!c->extra || !c->blocks || !c->mu) { pw_context_destroy(c); return PW_E_NOMEM; }
    {
        const char* ns = getenv("PW_SETS_IN_FLIGHT");
        c->nsets = ns ? atoi(ns) : 0;
        if (c->nsets != 0 && c->nsets < 2) c->nsets = 2;
        if (c->nsets > PW_SETS) c->nsets = PW_SETS;
        c->cur_sets = 2;
    }
    c->cur_queue = c->queue;
    {
        // the optimiser chains are the critical path: their launch gets the highest priority,
        // the average-diameter launch the lowest
        int lo = 0, hi = 0;
        CTX_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int b = 0; b < PW_SETS; ++b)
            CTX_TRY(hipStreamCreateWithPriority(&c->prods[b], hipStreamNonBlocking, hi));
        c->prod = c->prods[0];
        CTX_TRY(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, lo));
    }
    for (int b = 0; b < PW_SETS; ++b) {
        CTX_TRY(hipEventCreateWithFlags(&c->ev_reset[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_prod[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_gate[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_join[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_done[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_tail[b], hipEventDisableTiming));
        CTX_TRY(hipEventCreateWithFlags(&c->ev_head[b], hipEventDisableTiming));
        CTX_TRY(hipStreamCreateWithFlags(&c->cons[b], hipStreamNonBlocking));
    }
    {
        const char* hg = getenv("PW_HEAD_GATE");
        c->head_pct = hg ? atoi(hg) : -1;        // (-1: by batch size, gate_pct below)
        if (c->head_pct < -1 || c->head_pct > 100) c->head_pct = 0;
    }
    {
        const char* tg = getenv("PW_TAIL_GATE");
        c->tail_pct = tg ? atoi(tg) : -1;
        if (c->tail_pct < -1 || c->tail_pct > 100) c->tail_pct = 0;
    }
    c->need_fork = 1;
    ct_mark("streams+events");
    CTX_TRY(hipEventCreate(&c->ev0));
    CTX_TRY(hipEventCreate(&c->ev1));
    CTX_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CTX_TRY(hipEventCreateWithFlags(&c->ev_ext, hipEventDisableTiming));
    const char* fz = getenv("PW_FUSED");
    c->fused = (fz && fz[0] == '1') ? 1 : 0;
    c->c_waves = 4;
    c->prm = default_params();
    {
        auto ms_env = [](const char* name, long dflt) { const char* e = getenv(name); long v = e ? atol(e) : dflt; return v > 0 ? v : dflt; };
        c->wait_ticks = 100000ll * ms_env("PW_WAIT_LIMIT_MS", 250);          // (wall_clock64: 100 MHz)
        c->stream_wait_ticks = 100000ll * ms_env("PW_STREAM_LIMIT_MS", 5000);
        const char* tr = getenv("PW_TIMEOUT_REPEATS");
        c->timeout_repeats = tr ? atoi(tr) : 2;
        if (c->timeout_repeats < 0) c->timeout_repeats = 0;
    }
    if (!c->fused && !(getenv("PW_STREAM_PROBE") && getenv("PW_STREAM_PROBE")[0] == '0')) {
        // Do the ten streams of the pipeline really run side by side?  (PW_STREAM_PROBE=0 skips the question:
        // a profiler that serialises kernels -- rocprofv3 --pmc -- would otherwise turn the pipeline off and
        // count a different kernel.)  (GPU_MAX_HW_QUEUES is read when the
        // process initialises HIP: an application that did so before this library exported it -- PyTorch
        // imported and used first, a direct binding of the C ABI -- has the default of four hardware queues.)
        hipStream_t all[2 + 2 * PW_SETS];
        int k = 0;
        all[k++] = c->stream; all[k++] = c->aux;
        for (int b = 0; b < PW_SETS; ++b) { all[k++] = c->prods[b]; all[k++] = c->cons[b]; }
        int* flags = (int*)c->counter;                        // (the work counters are unused so far)
        // (a first round that waits for nobody: the code object is loaded and every stream has dispatched once
        // before the round that is timed against a 10 ms limit.  The flags are cleared on one of the streams
        // and the device is idle before any probe starts: these streams do not wait for the default stream.)
        CTX_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int), c->stream));
        CTX_TRY(hipDeviceSynchronize());
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(pw_probe_kernel, dim3(1), dim3(64), 0, all[i], flags, 0, flags + 1);
        CTX_TRY(hipDeviceSynchronize());
        // (up to three timed rounds: streams that share a hardware queue fail every round, but a host thread
        // that is not scheduled for 10 ms in the middle of the k launches - seen on loaded hosts,
        // tests/tools/launch_jitter.py - fails one, and the verdict lasts for the life of the context)
        int got[2] = {0, 0};
        for (int round = 0; round < 3 && got[1] < k; ++round) {
            CTX_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int), c->stream));
            CTX_TRY(hipDeviceSynchronize());
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(pw_probe_kernel, dim3(1), dim3(64), 0, all[i], flags, k, flags + 1);
            CTX_TRY(hipDeviceSynchronize());
            CTX_TRY(hipMemcpy(got, flags, sizeof(got), hipMemcpyDeviceToHost));
        }
        CTX_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int), c->stream));
        CTX_TRY(hipDeviceSynchronize());
        c->concurrent_streams = got[1];
        if (got[1] < k) {
            // not all of them: the overlapped pipeline cannot be trusted on this process -- every analysis
            // becomes ONE launch (all stages in a team), which needs no concurrency at all
            c->fused = 1;
            const char* q = getenv("GPU_MAX_HW_QUEUES");
            fprintf(stderr, "pywindow_amd: only %d of %d HIP streams run concurrently (GPU_MAX_HW_QUEUES=%s%s); analyses run as "
                            "single launches.  Export GPU_MAX_HW_QUEUES=12 before the process first touches HIP for the "
                            "overlapped pipeline (INTEGRATION.md).\n", got[1], k, q ? q : "unset",
                    q ? ", possibly exported after HIP was initialised" : "");
        }
    }
    ct_mark("stream-probe");
    {
        unsigned* host = new (std::nothrow) unsigned[65536];
        if (!host) { pw_context_destroy(c); return PW_E_NOMEM; }
        rsqrt14_decode(host);
        hipError_t e1 = hipMalloc((void**)&c->rsq_tab, 65536 * sizeof(unsigned));
        hipError_t e2 = e1 == hipSuccess
                            ? hipMemcpy(c->rsq_tab, host, 65536 * sizeof(unsigned), hipMemcpyHostToDevice)
                            : e1;
        delete[] host;
        if (e2 != hipSuccess) { set_err("rsqrt14 table upload", e2); pw_context_destroy(c); return PW_E_HIP; }
    }
    ct_mark("rsqrt-table");
    {
        // the neighbour tables of the sampling sphere: 2.1 M rows of 40 bytes, built once (a few ms)
        const char* nb = getenv("PW_NB_TABLES");
        if (!(nb && nb[0] == '0')) {
            const size_t rows = (size_t)nb_dense_offset(PW_NB_PMAX + 1);
            CTX_TRY(hipMalloc((void**)&c->nb_off, (PW_NB_PMAX + 1) * sizeof(unsigned)));
            CTX_TRY(hipMemset(c->nb_off, 0xff, (PW_NB_PMAX + 1) * sizeof(unsigned)));
            CTX_TRY(hipMalloc((void**)&c->nb_idx, rows * PW_NB_K * sizeof(unsigned short)));
            CTX_TRY(hipMalloc((void**)&c->nb_bound, rows * sizeof(double)));
            const char* nu = getenv("PW_UNIT_TABLE");
            if (!(nu && nu[0] == '0')) CTX_TRY(hipMalloc((void**)&c->nb_unit, rows * 3 * sizeof(double)));
            hipLaunchKernelGGL(pw_nb_build_kernel, dim3(PW_NB_PMAX - PW_NB_PMIN + 1), dim3(256), 0, c->stream, c->nb_off,
                               c->nb_idx, c->nb_bound, c->nb_unit);
            CTX_TRY(hipGetLastError());
            // (no wait: the kernel takes 8 ms and the first thing to read the tables is an analysis launch, which is
            // ordered behind everything on the API stream -- need_fork below, and single launches run on that stream)
        }
    }
#undef CTX_TRY
    ct_mark("neighbour-tables");
    if (ctx_timing) fprintf(stderr, "pywindow_amd: pw_context_create, ms by leg:%s\n", ct_text);
    *out = c;
    return PW_OK;
}

void pw_context_destroy(pw_context* c) {
    if (!c) return;
    if (c->device < 0) { delete c->extra; delete c->mu; delete c; return; }
    DeviceScope scope;
    (void)scope.enter(c->device);
    (void)hipDeviceSynchronize();
    if (c->ws) (void)hipFree(c->ws);
    if (c->slab) (void)hipFree(c->slab);
    if (c->blocks) for (auto& b : *c->blocks) (void)hipFree(b.p);
    delete c->blocks;
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->ready_vals) (void)hipHostFree(c->ready_vals);
    if (c->bigmem) (void)hipFree(c->bigmem);
    delete c->extra;
    if (c->ev_ext) (void)hipEventDestroy(c->ev_ext);
    for (int k = 0; k < 3; ++k)
        for (int e = 0; e < 2; ++e)
            if (c->ev_t[k][e]) (void)hipEventDestroy(c->ev_t[k][e]);
    if (c->counter) (void)hipFree(c->counter);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int b = 0; b < PW_SETS; ++b) {
        if (c->ev_reset[b]) (void)hipEventDestroy(c->ev_reset[b]);
        if (c->ev_prod[b]) (void)hipEventDestroy(c->ev_prod[b]);
        if (c->ev_gate[b]) (void)hipEventDestroy(c->ev_gate[b]);
        if (c->ev_join[b]) (void)hipEventDestroy(c->ev_join[b]);
        if (c->ev_done[b]) (void)hipEventDestroy(c->ev_done[b]);
        if (c->ev_tail[b]) (void)hipEventDestroy(c->ev_tail[b]);
        if (c->ev_head[b]) (void)hipEventDestroy(c->ev_head[b]);
        if (c->cons[b]) (void)hipStreamDestroy(c->cons[b]);
    }
    if (c->adj) (void)hipFree(c->adj);
    if (c->pool) (void)hipFree(c->pool);
    if (c->queue) (void)hipFree(c->queue);
    if (c->rsq_tab) (void)hipFree(c->rsq_tab);
    if (c->nb_off) (void)hipFree(c->nb_off);
    if (c->nb_idx) (void)hipFree(c->nb_idx);
    if (c->nb_bound) (void)hipFree(c->nb_bound);
    if (c->nb_unit) (void)hipFree(c->nb_unit);
    if (c->slots) (void)hipFree(c->slots);
    for (int b = 0; b < PW_SETS; ++b)
        if (c->prods[b]) (void)hipStreamDestroy(c->prods[b]);
    if (c->rb_stream) (void)hipStreamDestroy(c->rb_stream);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c->mu;
    delete c;
}

void pw_params_default(pw_params* p) {
    if (p) *p = default_params();
}

int pw_context_set_params(pw_context* c, const pw_params* p) {
    if (!c || !p) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (!(p->adjust_windows > 0.0) || !(p->adjust_average > 0.0) || !(p->increment > 0.0) || !(p->increment2 > 0.0)) {
        snprintf(g_err, sizeof(g_err), "pw_params: adjust and increment must be positive");
        return PW_E_BAD_ARG;
    }
    if (p->opt_flags & PW_OPT_CUSTOM_BOUNDS)
        for (int k = 0; k < 3; ++k)
            if (p->opt_lo[k] > p->opt_hi[k]) {   // scipy raises for the same input
                snprintf(g_err, sizeof(g_err), "pw_params: an upper bound is less than the corresponding lower bound");
                return PW_E_BAD_ARG;
            }
    if (!p->lb_z && p->z_lo > p->z_hi) {
        snprintf(g_err, sizeof(g_err), "pw_params: an upper bound is less than the corresponding lower bound");
        return PW_E_BAD_ARG;
    }
    c->prm = *p;
    c->p_cap_min = 0;        // (what an earlier batch asked for beyond its knobs does not outlive them)
    c->prm.pore_opt = p->pore_opt ? 1 : 0;
    c->prm.lb_z = p->lb_z ? 1 : 0;
    c->prm.z_second_mini = p->z_second_mini ? 1 : 0;
    return PW_OK;
}

int pw_context_device(pw_context* c) { return c ? c->device : -1; }

// Device scratch owned by the context and kept between calls: the periodic re-assembly needs a few
// megabytes per team (gigabytes per launch), and allocating and freeing that around every call cost more
// than the launch itself -- and made its time depend on the state of the allocator.  Work that uses
// the buffer is ordered on the API stream; growing it waits for the device.
int pw_internal_pool(pw_context* c, size_t bytes, void** out) {
    if (!c || !out) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "device scratch");
    if (c->pool_bytes < bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->pool) HIP_TRY(hipFree(c->pool));
        c->pool = nullptr;
        c->pool_bytes = 0;
        HIP_TRY(hipMalloc(&c->pool, bytes));
        c->pool_bytes = bytes;
    }
    *out = c->pool;
    return PW_OK;
}
char* pw_internal_error_buffer(void) { return g_err; }

// The stream of the periodic re-assembly (pw_rebuild.hip).  A trajectory goes through in pieces, and the re-assembly of
// piece k + 1 runs while piece k is being analysed: on the API stream (normal priority) its teams -- 74 KB of LDS each --
// stood in line behind the analysis' own and a launch that takes 1.5 ms alone took 5.6 (profiles/r06_periodic_*), with the
// next analysis waiting for it.  On a stream of the highest priority its workgroups are placed first.  Every call on it
// ends with a host synchronisation, so nothing else has to be ordered against it.  PW_RB_STREAM=0: the API stream.
void* pw_internal_rebuild_stream(pw_context* c) {
    if (!c || c->device < 0) return nullptr;
    const char* e = getenv("PW_RB_STREAM");
    if (e && e[0] == '0') return (void*)c->stream;
    if (!c->rb_stream) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
            hipStreamCreateWithPriority(&c->rb_stream, hipStreamNonBlocking, hi) != hipSuccess) {
            c->rb_stream = nullptr;
            return (void*)c->stream;
        }
    }
    return (void*)c->rb_stream;
}

void* pw_context_stream(pw_context* c) { return c ? (void*)c->stream : nullptr; }

// Page-locked host staging buffer of the context (grown on demand, one per context): the reader decodes
// frames straight into it and pw_resident_upload's copies from it are real asynchronous DMA instead of
// going through the runtime's bounce buffer.  Valid until the next call that asks for more.
int pw_context_pinned(pw_context* c, size_t bytes, void** out) {
    if (!c || !out) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "pinned staging");
    PW_ON_DEVICE(c->device);
    if (c->pinned_bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->pinned) HIP_TRY(hipHostFree(c->pinned));
        c->pinned = nullptr;
        c->pinned_bytes = 0;
        const size_t want = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
        HIP_TRY(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
        c->pinned_bytes = want;
    }
    *out = c->pinned;
    return PW_OK;
}

// number of host threads of a device = -1 context (0: keep); returns the current number
int pw_context_host_threads(pw_context* c, int threads) {
    if (!c || c->device >= 0) return 0;
    PW_LOCK_CONTEXT(c);
    if (threads > 0) c->host_threads = threads;
    return c->host_threads;
}

int pw_resident_launch(pw_context* c, pw_resident* r, uint32_t stages) {
    if (!c || !r) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (r->n_units == 0) return PW_OK;
    if (c->device < 0) {
        if (!r->host) return PW_E_BAD_ARG;
        pw_batch_in in = {r->n_units, r->host->offset.data(), r->host->xyz.data(), r->host->vdw.data(), r->host->mass.data(),
                          r->host->template_atoms};
        return host_analyse(c, &in, stages, r->host->out.data(), nullptr);
    }
    PW_ON_DEVICE(c->device);
    stages &= PW_STAGE_ALL;
    int rc;
    if (r->d_ready && r->ready_units < r->n_units) {
        // A streamed batch whose coordinates are still arriving.  The pipeline's launches run on streams of their own and
        // wait for the units they are handed; a single launch (a context without the pipeline: PW_FUSED=1, hardware
        // queues taken by whoever initialised the GPU first -- or stages without the window search) would sit on the
        // API stream in front of the very copies it waits for.  It is made when the last unit has arrived.
        const bool large_ = wanted_p_cap(c) > PW_P_CAP_PIPELINE;
        if (c->fused || (stages & PW_STAGE_WINDOWS) == 0 || large_) {
            r->pending = 1;
            r->pending_stages = stages;
            return PW_OK;
        }
    }
    r->pending = 0;
    {
        // a molecule whose coordinates do not fit a CU's LDS (about 1700 atoms): the same source with the
        // team's shared block in global memory (pw_kernels_big.hip), one launch, every stage in a team
        const int pcap = wanted_p_cap(c) > c->p_cap ? wanted_p_cap(c) : c->p_cap;
        const bool win_ = (stages & PW_STAGE_WINDOWS) != 0, opt_ = win_ || (stages & PW_STAGE_OPT);
        // (the smallest team either launch shape can fall back to: one wave, one window-fit slot)
        const size_t need = (!c->fused && win_) ? UnitShared::bytes(r->nmax, 1, 1, 1, false, pcap)
                                                : UnitShared::bytes(r->nmax, win_ ? 1 : 0, win_ ? 1 : (opt_ ? 1 : 0), 2, false, pcap);
        if (need + 64 > 160 * 1024 - 256 - PW_KERNEL_STATIC_LDS) {
            if (r->nmax > 40000) {
                snprintf(g_err, sizeof(g_err), "molecule with %d atoms: more than the 40000 the pair indices are sized for", r->nmax);
                return PW_E_TOO_LARGE;
            }
            long grid = 2L * c->n_cu;
            if (grid > r->n_units) grid = r->n_units;
            rc = ensure_workspace(c, (int)grid, (int)grid);
            if (rc != PW_OK) return rc;
            const size_t bb = pw_internal_big_block_bytes(r->nmax, c->p_cap);
            if (c->bigmem_bytes < bb * (size_t)grid) {
                HIP_TRY(hipDeviceSynchronize());
                if (c->bigmem) HIP_TRY(hipFree(c->bigmem));
                c->bigmem = nullptr;
                c->bigmem_bytes = 0;
                HIP_TRY(hipMalloc((void**)&c->bigmem, bb * (size_t)grid));
                c->bigmem_bytes = bb * (size_t)grid;
            }
            c->need_fork = 1;
            rc = join_pipeline(c);
            if (rc != PW_OK) return rc;
            if (r->read_valid[r->cur]) {
                HIP_TRY(hipStreamWaitEvent(c->stream, r->ev_read[r->cur], 0));
                r->read_valid[r->cur] = 0;
            }
            r->written_set[r->cur] = -1;
            HIP_TRY(hipMemsetAsync(r->d_xw_count + r->cur, 0, sizeof(unsigned), c->stream));
            HIP_TRY(hipMemsetAsync(c->counter + 4 * PW_SETS + 1, 0, sizeof(unsigned long long), c->stream));
            PwWsArgs wsa;
            wsa.ws = c->ws; wsa.slab = c->slab; wsa.adj = c->adj;
            wsa.xwin = r->d_xw[r->cur]; wsa.xwin_count = r->d_xw_count + r->cur; wsa.xwin_cap = r->xw_cap;
            wsa.p_cap = c->p_cap;
            wsa.nb_off = c->nb_off; wsa.nb_idx = c->nb_idx; wsa.nb_bound = c->nb_bound; wsa.nb_unit = c->nb_unit;
            wsa.wait_ticks = c->wait_ticks; wsa.stream_wait_ticks = c->stream_wait_ticks;
            return pw_internal_big_launch((void*)c->stream, (int)grid, r->n_units, r->d_offset, r->d_xyz, r->d_vdw, r->d_mass,
                                          stages, r->nmax, &wsa, c->bigmem, bb, c->counter + 4 * PW_SETS + 1, r->d_out, &c->prm,
                                          c->rsq_tab, r->vstride);
        }
    }
    const bool large = wanted_p_cap(c) > PW_P_CAP_PIPELINE;
    const bool pipeline = !c->fused && (stages & PW_STAGE_WINDOWS) != 0 && !large;
    if (!pipeline) {
        // one launch: every requested stage inside the same team
        LaunchPlan p;
        bool win = (stages & PW_STAGE_WINDOWS) != 0;
        bool opt = win || (stages & PW_STAGE_OPT);
        rc = plan_launch(c, r->n_units, r->nmax, (win || (stages & PW_STAGE_AVG)) ? 4 : 1, win,
                         win ? -1 : (opt ? 1 : 0), &p);
        if (rc != PW_OK) return rc;
        if (large) {
            // tens of thousands of sampling vectors per molecule (adjust beyond 4): as many teams as 48 GB of
            // workspace allow -- the adjacency rows of DBSCAN are p_cap^2 / 8 bytes per team
            const size_t per = team_ws_bytes(wanted_p_cap(c), win);
            const size_t budget = (size_t)48 << 30;
            if (per > ((size_t)192 << 30)) {
                snprintf(g_err, sizeof(g_err), "%d sampling vectors per molecule need %zu GB of workspace per team",
                         wanted_p_cap(c), per >> 30);
                return PW_E_TOO_LARGE;
            }
            long g = (long)(budget / per);
            if (g < 1) g = 1;
            if (g < p.grid) p.grid = (int)g;
        }
        rc = ensure_workspace(c, p.grid, win ? p.grid : 0);
        if (rc != PW_OK) return rc;
        c->need_fork = 1;
        rc = join_pipeline(c);
        if (rc != PW_OK) return rc;
        if (r->read_valid[r->cur]) {
            HIP_TRY(hipStreamWaitEvent(c->stream, r->ev_read[r->cur], 0));
            r->read_valid[r->cur] = 0;
        }
        r->written_set[r->cur] = -1;       // (the API stream joins every pipeline launch: nothing to remember)
        HIP_TRY(hipMemsetAsync(r->d_xw_count + r->cur, 0, sizeof(unsigned), c->stream));
        return launch_plan(c, r, stages, p, c->stream, 0, win ? 0 : -1, 4 * PW_SETS);
    }
    // Pipeline: the analysis is split by parallel shape and the pieces overlap.
    //   A (producer stream): stage_basic + pore-centre optimiser, ONE wave per unit -- the serial
    //     chain; every unit of a 1000-frame batch iterates at once.  Chains differ 10x in length;
    //     each finished unit is published to a queue.
    //   B (aux stream): average diameter, 4 waves per unit, independent of A.
    //   C (the set's consumer stream): window search, persistent teams consuming units as A publishes them.
    //     A one-wave gate kernel ahead of C (and B) holds them back until every team of A is
    //     resident, so they can never take the LDS A needs -- no launch-order assumption.
    LaunchPlan pa, pb, pc;
    rc = plan_launch(c, r->n_units, r->nmax, 1, false, 1, &pa, 1, true);   // chains: no shifted frame
    if (rc != PW_OK) return rc;
    pa.grid = (int)r->n_units < pa.grid ? (int)r->n_units : pa.grid;
    {
        // PW_A_LDS_KB: pad the LDS request of the optimiser teams (tuning: keeps the bulk launches
        // off a CU while its chains are young)
        const char* al = getenv("PW_A_LDS_KB");
        if (al && atoi(al) > 0) {
            size_t want = (size_t)atoi(al) * 1024;
            if (want > pa.lds && want <= 160 * 1024 - 256 - PW_KERNEL_STATIC_LDS) pa.lds = want;
        }
        if (getenv("PW_PLAN_DEBUG")) fprintf(stderr, "plan A: grid %d lds %zu\n", pa.grid, pa.lds);
    }
    bool do_avg = (stages & PW_STAGE_AVG) != 0;
    // Round 6: the average diameter is a stage of the WINDOW teams (before the window search of the unit they have just
    // taken), not a launch of its own.  Its teams -- four waves, one per SIMD of a CU, 96 of them persistent -- took the
    // four wave slots of their CU that the optimiser chains live on (two 256-register waves fill a SIMD, and a
    // 168-register wave beside one leaves no room for a second), i.e. 384 of the chip's 1024 chain slots, busy or not;
    // as a stage of the window teams the same work needs no slots of its own and a fifth more window teams fit
    // (1000 units 1.16 -> 1.13 ms per step, 4000 units 4.41 -> 4.13 ms, 500 units 0.91 -> 0.64:
    // profiles/r06_avg_in_window_teams.txt).  PW_B_LAUNCH=1: the separate launch of rounds 1-5.
    bool avg_in_c = false;
    if (do_avg && !(getenv("PW_B_LAUNCH") && getenv("PW_B_LAUNCH")[0] == '1')) { avg_in_c = true; do_avg = false; }
    pb.grid = 0;
    if (do_avg) {
        // one frame; the optimiser-state slots are this launch's scratch arena: 7 (51 KB) hold the ray vectors and
        // the cone pairs of team_ray_tests, 2 (14 KB) make the stage fall back to the dense ray scan (PW_B_LB)
        int b_lb = 7;
        if (const char* e = getenv("PW_B_LB")) b_lb = atoi(e) > 0 ? atoi(e) : b_lb;
        rc = plan_launch(c, r->n_units, r->nmax, 4, false, b_lb, &pb, 1, true);
        if (rc != PW_OK) return rc;
    }
    // The window search: ONE launch of 4-wave teams, sampling and fits (the split into a sampling launch and one-wave
    // fit workers that round 4 built was 4-25x slower and is gone: profiles/r04_split_*)
    rc = plan_launch(c, r->n_units, r->nmax, c->c_waves, true, -1, &pc, 1);      // one frame, shifted in place
    if (rc != PW_OK) return rc;
    // A batch of up to a few units per SIMD is latency-bound by its optimiser chains: one window team
    // per CU keeps LDS free for the chains of the next launch (measured on 1000 units: 2.56 -> 2.45 ms);
    // larger batches want every team the LDS admits (4000 units: 9.2 ms against 10.0).
    // (round 5, after the optimiser chains got a third faster: beyond that, five teams per four CUs -- 4000 units 4.80 ms
    // against 5.14 with two per CU, 8192: 9.41 / 10.1, 20 000: 22.8 / 24.3; profiles/r05_resweep.txt)
    if (avg_in_c) {
        // (with the average-diameter teams gone: five window teams per four CUs up to 1500 units -- 1000 units 1.126 ms with
        // 320 teams, 1.163 with 304, 1.187 with 288, 1.127 with 352 -- eleven per eight CUs beyond: 4000 units 4.13 ms with
        // 352 teams, 4.27 with 320, 4.31 with 384; small batches one team per two units)
        const long cap = r->n_units <= 6L * c->n_cu ? c->n_cu + c->n_cu / 4 : c->n_cu + (3 * c->n_cu) / 8;
        if (pc.grid > cap) pc.grid = (int)cap;
        if (r->n_units <= 300 && pc.grid > (r->n_units + 1) / 2) pc.grid = (int)((r->n_units + 1) / 2);
    } else if (r->n_units <= 6L * c->n_cu && pc.grid > c->n_cu) pc.grid = c->n_cu;
    else if (pc.grid > c->n_cu + c->n_cu / 4) pc.grid = c->n_cu + c->n_cu / 4;
    // ... and the average-diameter launch, a fifth of the window search's work, gets by with one team
    // per two CUs whatever the batch (1000 units: 1.84 -> 1.79 ms, 500: 1.28 -> 1.20; 4000: 6.59 -> 6.53)
    const int pb_planned = pb.grid;
    // (round 5: three teams per eight CUs -- 1000 units 1.25-1.27 ms against 1.29 with one per two CUs, 4000 the same)
    if (do_avg && pb.grid > (3 * c->n_cu + 7) / 8) pb.grid = (3 * c->n_cu + 7) / 8;
    {
        // PW_C_TEAMS / PW_B_TEAMS: cap the persistent teams of the window / average launches (tuning)
        const char* ct = getenv("PW_C_TEAMS");
        if (ct && atoi(ct) > 0) {
            long g = (long)c->n_cu * 4;
            if (g > r->n_units) g = r->n_units;
            int per_cu = (int)(c->lds_per_cu / pc.lds);
            if (per_cu < 1) per_cu = 1;
            if (g > (long)c->n_cu * per_cu) g = (long)c->n_cu * per_cu;
            pc.grid = atoi(ct) < g ? atoi(ct) : (int)g;
        }
        const char* bt = getenv("PW_B_TEAMS");
        if (bt && do_avg && atoi(bt) > 0) pb.grid = atoi(bt) < pb_planned ? atoi(bt) : pb_planned;   // (may also raise it)
    }
    if (const char* cslots = getenv("PW_C_SLOTS")) {
        // experiment: fewer window-fit slots than waves (less LDS per team, windows fitted in rounds);
        // PW_C_TEAMS then sets the number of teams (up to what the smaller request admits per CU)
        int k = atoi(cslots);
        if (k >= 1 && k < 4 && pc.nw == 4) {
            pc.nrot = pc.nlb = k;
            pc.lds = UnitShared::bytes(r->nmax, k, k, 1, false, wanted_p_cap(c) > c->p_cap ? wanted_p_cap(c) : c->p_cap) + 64;
            int per_cu = (int)(c->lds_per_cu / pc.lds);
            if (per_cu > 4) per_cu = 4;
            long g = (long)c->n_cu * per_cu;
            const char* ct = getenv("PW_C_TEAMS");
            if (ct && atoi(ct) > 0 && atoi(ct) < g) g = atoi(ct);
            pc.grid = (int)(g < r->n_units ? g : r->n_units);
            if (getenv("PW_PLAN_DEBUG")) fprintf(stderr, "plan C: slots %d lds %zu grid %d\n", k, pc.lds, pc.grid);
        }
    }
    // teams: C[0..n) | A[0..n) | B (n = sets in flight: that many window and optimiser launches can be
    // running at once; the average-diameter launches follow each other on one stream), every region
    // as large as the largest grid seen so far -- see pw_context::max_a
    // How many analyses may be in flight.  A small batch leaves most of the chip idle while its few long
    // optimiser chains finish (the slowest of 1000 takes 3.4 ms, the mean 1 ms), so the period of
    // back-to-back analyses is their latency divided by the number in flight until the window teams
    // saturate: measured on MI355X, 125 frames: 2.13 ms with two sets, 1.19 with four; 1000 frames:
    // 2.29 -> 2.22.  Large batches fill the chip on their own and only pay for the extra workspaces.
    // (More than four would need more than 2 + 2 x 4 streams; beyond the hardware queues the runtime
    // grants, streams share queues and a gate kernel then blocks the launch it is waiting for until its
    // time-out -- measured: 8 sets, 2 s per step.)
    int ns = c->nsets;
    // (round 4, with the optimiser chains a quarter faster: 1000 frames 1.57 ms with four sets, 1.54-1.55 with three;
    // 250 frames 0.86 against 1.07; 4000 frames 5.96 / 5.84 / 5.80 with four / three / two)
    if (ns == 0) ns = auto_sets(r->n_units);
    if (ns > r->nbuf) ns = r->nbuf;
    if (pa.grid > c->max_a || pb.grid > c->max_b || pc.grid > c->max_c || ns != c->cur_sets) {
        HIP_TRY(hipDeviceSynchronize());         // nothing is in flight while the layout changes
        if (pa.grid > c->max_a) c->max_a = pa.grid;
        if (pb.grid > c->max_b) c->max_b = pb.grid;
        if (pc.grid > c->max_c) c->max_c = pc.grid;
        if (ns != c->cur_sets) {
            c->cur_sets = ns;
            c->flip = -1;
            for (int k = 0; k < PW_SETS; ++k) c->done_valid[k] = c->tail_valid[k] = c->head_valid[k] = 0;
        }
    }
    const int ws_b = ns * (c->max_c + c->max_a);
    rc = ensure_workspace(c, ws_b + c->max_b, ns * c->max_c);
    if (rc != PW_OK) return rc;
    if (c->slots_cap < r->n_units) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->slots) HIP_TRY(hipFree(c->slots));
        c->slots = nullptr;
        HIP_TRY(hipMalloc((void**)&c->slots, PW_SETS * sizeof(int) * (size_t)r->n_units));
        c->slots_cap = r->n_units;
        for (int k = 0; k < PW_SETS; ++k) c->done_valid[k] = c->tail_valid[k] = c->head_valid[k] = 0;
    }
    // this launch's set; the others may still be in use by the launches before it.  p: the set of the
    // launch just before this one (its queue is what this launch's gates watch); nx: the set after
    // this one, i.e. the launch that watched the queue this launch is about to reset
    const int p = c->flip;
    const int b = (c->flip + 1) % ns;
    const int nx = (b + 1) % ns;
    c->flip = b;
    const int ws_a = ns * c->max_c + b * c->max_a, ws_c = b * c->max_c;
    {
        const char* ps = getenv("PW_PROD_STREAMS");
        c->prod = c->prods[((ps && ps[0] == '2') || c->tail_pct != 0) ? b : 0];
    }
    r->cur = (r->cur + 1) % r->nbuf;
    r->d_out = r->d_outs[r->cur];
    c->cur_queue = c->queue + b;
    c->cur_slots = c->slots + (size_t)b * c->slots_cap;
    if (c->need_fork) {
        // uploads, single-launch analyses and timing marks on the main stream come first
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        for (int k = 0; k < PW_SETS; ++k) {
            HIP_TRY(hipStreamWaitEvent(c->prods[k], c->ev_fork, 0));
            HIP_TRY(hipStreamWaitEvent(c->cons[k], c->ev_fork, 0));
        }
        HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        c->need_fork = 0;
    }
    // the launch that used this set before has finished (it is `ns` launches back) ...
    if (c->done_valid[b]) HIP_TRY(hipStreamWaitEvent(c->prod, c->ev_done[b], 0));
    // ... and so has the launch that last wrote this result buffer (the same one unless several
    // batches alternate on the context)
    if (r->written_set[r->cur] >= 0 && r->written_set[r->cur] != b && c->done_valid[r->written_set[r->cur]])
        HIP_TRY(hipStreamWaitEvent(c->prod, c->ev_done[r->written_set[r->cur]], 0));
    r->written_set[r->cur] = b;
    if (r->read_valid[r->cur]) {     // a gather on the caller's stream may still be reading this buffer
        HIP_TRY(hipStreamWaitEvent(c->prod, r->ev_read[r->cur], 0));
        r->read_valid[r->cur] = 0;
    }
    // the gates of the launch after this set's previous user read the queue that is reset below
    if (c->tail_valid[nx]) HIP_TRY(hipStreamWaitEvent(c->prod, c->ev_tail[nx], 0));
    if (c->head_valid[nx]) HIP_TRY(hipStreamWaitEvent(c->prod, c->ev_head[nx], 0));
    {
        static_assert(sizeof(pw_unit_out) % 8 == 0, "records are cleared in 8-byte words");
        const long n8 = (long)(sizeof(pw_unit_out) / 8) * r->n_units;
        long blocks = (n8 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pw_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, c->prod, (unsigned long long*)r->d_out, n8,
                           c->cur_queue, c->cur_slots, r->n_units, c->counter + b, c->counter + PW_SETS + b,
                           c->counter + 2 * PW_SETS + b, c->counter + 3 * PW_SETS + b, r->d_xw_count + r->cur);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(c->ev_reset[b], c->prod));
    c->tail_valid[b] = 0;
    // The gates pace RE-launches of one batch (the steps of a streamed analysis: start where the older launch
    // leaves the chip idle).  Launches of DIFFERENT batches -- the pieces of one trajectory on their way
    // through -- are independent work that should all be in flight as soon as it arrives: no gates, no
    // ordering against the previous launch.
    const bool same_batch = p >= 0 && c->last_res[p] == (const void*)r;
    const bool have_prev = p >= 0 && p != b && c->done_valid[p] && c->last_units[p] > 0 && same_batch;
    const int tail_pct = gate_pct(c->tail_pct, r->n_units), head_pct = gate_pct(c->head_pct, r->n_units);
    if (tail_pct > 0 && have_prev) {
        unsigned long long need = (unsigned long long)((c->last_units[p] * tail_pct) / 100);
        hipLaunchKernelGGL(pw_tail_gate_kernel, dim3(1), dim3(64), 0, c->prod, c->queue + p, need, c->counter + 4 * PW_SETS + 2);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_tail[b], c->prod));
        c->tail_valid[b] = 1;
    }
    c->last_units[b] = r->n_units;
    c->last_res[b] = (const void*)r;
    // several optimiser launches can be in flight: separate work counters and workspaces
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t[0][0], c->prod));
    rc = launch_plan(c, r, PW_STAGE_BASIC | PW_STAGE_OPT | PW_STAGE_MERGE, pa, c->prod, ws_a, -1, b, PW_ROLE_PRODUCER, false);
    if (rc != PW_OK) return rc;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t[0][1], c->prod));
    HIP_TRY(hipEventRecord(c->ev_prod[b], c->prod));
    hipStream_t cs = c->cons[b];
    HIP_TRY(hipStreamWaitEvent(cs, c->ev_reset[b], 0));
    hipLaunchKernelGGL(pw_gate_kernel, dim3(1), dim3(64), 0, cs, c->cur_queue,
                       (long)pa.grid < r->n_units ? pa.grid : (int)r->n_units, c->counter + 4 * PW_SETS + 2, c->wait_ticks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_gate[b], cs));
    c->head_valid[b] = 0;
    if (have_prev) {
        if (head_pct > 0) {
            // start beside the tail of the previous window launch, not behind it
            unsigned long long need_h = (unsigned long long)((c->last_units[p] * head_pct) / 100);
            hipLaunchKernelGGL(pw_head_gate_kernel, dim3(1), dim3(64), 0, cs, c->queue + p, need_h, c->counter + 4 * PW_SETS + 2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(c->ev_head[b], cs));
            c->head_valid[b] = 1;
        } else {
            HIP_TRY(hipStreamWaitEvent(cs, c->ev_done[p], 0));
        }
    }
    if (c->timing) { c->timed_avg = do_avg ? 1 : 0; HIP_TRY(hipEventRecord(c->ev_t[2][0], cs)); }
    rc = launch_plan(c, r, avg_in_c ? MASK_WINDOWS_AVG : MASK_WINDOWS, pc, cs, ws_c, b * c->max_c, PW_SETS + b, PW_ROLE_CONSUMER, false);
    if (rc != PW_OK) return rc;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev_t[2][1], cs));
    if (do_avg) {
        HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_gate[b], 0));
        if (c->timing) HIP_TRY(hipEventRecord(c->ev_t[1][0], c->aux));
        // (the launch's own counter: the previous average-diameter launch may still be running)
        rc = launch_plan(c, r, PW_STAGE_AVG | PW_STAGE_MERGE | PW_STAGE_COM_ONLY, pb, c->aux, ws_b, -1, 2 * PW_SETS + b,
                         PW_ROLE_PLAIN, false);
        if (rc != PW_OK) return rc;
        if (c->timing) HIP_TRY(hipEventRecord(c->ev_t[1][1], c->aux));
        HIP_TRY(hipEventRecord(c->ev_join[b], c->aux));
    }
    HIP_TRY(hipStreamWaitEvent(cs, c->ev_prod[b], 0));
    if (do_avg) HIP_TRY(hipStreamWaitEvent(cs, c->ev_join[b], 0));
    HIP_TRY(hipEventRecord(c->ev_done[b], cs));
    c->done_valid[b] = 1;
    return PW_OK;
}

// a consumer team that gave up waiting for the optimiser launch sets its queue's error flag: every
// call that waits for results reports it (and clears it), so a timed-out analysis is never mistaken
// for a finished one -- neither downloaded nor timed
static int check_queue_error(pw_context* c) {
    UnitQueue q[PW_SETS];
    HIP_TRY(hipMemcpy(q, c->queue, sizeof(q), hipMemcpyDeviceToHost));
    int which = -1;
    for (int b = 0; b < PW_SETS; ++b)
        if (q[b].error != 0 && which < 0) which = b;
    if (which >= 0) {
        // (only the flags: the other sets' queues may belong to analyses that are running)
        for (int b = 0; b < PW_SETS; ++b)
            if (q[b].error != 0) (void)hipMemset(&c->queue[b].error, 0, sizeof(c->queue[b].error));
        // cause 1: a window team saw no unit published by the optimiser launch for a whole limit (PW_WAIT_LIMIT_MS, 250 ms);
        // 2: a team saw no coordinates appended to a streamed batch for PW_STREAM_LIMIT_MS (5 s); 3: a streamed batch was given
        // up while its launches were waiting; 4: NOT ONE optimiser team of a launch started within the limit (pw_gate_kernel)
        const int cause = q[which].error;
        snprintf(g_err, sizeof(g_err), "%s (set %d of %d, cause %d: %llu units published, %llu taken, %d optimiser teams started)",
                 cause == 1 ? "window launch timed out waiting for the optimiser launch"
                            : (cause == 2 ? "a launch timed out waiting for the coordinates of a streamed batch"
                                          : (cause == 4 ? "the optimiser launch timed out becoming resident (residency gate)"
                                                        : "a streamed batch was given up while it was being analysed")),
                 which, c->cur_sets, cause, q[which].tail, q[which].head, q[which].started);
        // (a batch that was given up is not a time-out: nothing to repeat, nothing to count)
        return cause == 3 ? PW_E_HIP : PW_E_TIMEOUT;
    }
    return PW_OK;
}

int pw_resident_sync(pw_context* c) {
    if (!c) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (c->device < 0) return PW_OK;
    PW_ON_DEVICE(c->device);
    int rcj = join_pipeline(c);
    if (rcj != PW_OK) return rcj;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return check_queue_error(c);
}

int pw_resident_upload(pw_context* c, const pw_batch_in* in, pw_resident** out) {
    if (!c || !in || !out || in->n_units < 0 || in->template_atoms < 0) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    *out = nullptr;
    if (c->device < 0) {
        pw_resident* h = new (std::nothrow) pw_resident();
        if (!h) return PW_E_NOMEM;
        memset((void*)h, 0, sizeof(*h));
        h->host = new (std::nothrow) HostBatch();
        if (!h->host) { delete h; return PW_E_NOMEM; }
        const int64_t na = in->n_units ? in->atom_offset[in->n_units] : 0;
        const int64_t nc = in->template_atoms > 0 ? in->template_atoms : na;
        h->n_units = (long)in->n_units;
        h->n_atoms = (long)na;
        h->host->offset.assign(in->atom_offset, in->atom_offset + in->n_units + 1);
        h->host->xyz.assign(in->xyz, in->xyz + 3 * na);
        h->host->vdw.assign(in->vdw, in->vdw + nc);
        h->host->mass.assign(in->mass, in->mass + nc);
        h->host->out.resize((size_t)in->n_units);
        h->host->template_atoms = in->template_atoms;
        *out = h;
        return PW_OK;
    }
    PW_ON_DEVICE(c->device);
    pw_resident* r = new (std::nothrow) pw_resident();
    if (!r) return PW_E_NOMEM;
    memset(r, 0, sizeof(*r));
    r->n_units = (long)in->n_units;
    long natoms = in->n_units ? (long)in->atom_offset[in->n_units] : 0;
    r->n_atoms = natoms;
    int nmax = 0;
    for (long u = 0; u < r->n_units; ++u) {
        long n = (long)(in->atom_offset[u + 1] - in->atom_offset[u]);
        if (n <= 0 || (in->template_atoms > 0 && n != in->template_atoms)) {
            delete r;
            if (n <= 0) snprintf(g_err, sizeof(g_err), "unit %ld has %ld atoms", u, n);
            else snprintf(g_err, sizeof(g_err), "unit %ld has %ld atoms, the template %ld", u, n, (long)in->template_atoms);
            return PW_E_BAD_ARG;
        }
        if (n > nmax) nmax = (int)n;
    }
    r->nmax = nmax;
    r->vstride = in->template_atoms > 0 ? 0 : 1;
    // radii and masses: per atom, or ONE template for a batch of one molecule type (a trajectory)
    const long nconst = in->template_atoms > 0 ? (long)in->template_atoms : natoms;
#define UP_TRY(call)                          \
    do {                                      \
        hipError_t e_ = (call);               \
        if (e_ != hipSuccess) {               \
            set_err(#call, e_);               \
            pw_resident_free(c, r);           \
            return PW_E_HIP;                  \
        }                                     \
    } while (0)
    if (r->n_units) {
        r->nbuf = c->nsets ? c->nsets : auto_sets(r->n_units);
        // ONE device block for the whole batch, taken from the context's cache of freed blocks
        auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t b_off = up256(sizeof(long) * (size_t)(r->n_units + 1)), b_xyz = up256(sizeof(double) * 3 * (size_t)natoms);
        const size_t b_con = up256(sizeof(double) * (size_t)nconst);
        const size_t b_out = up256((size_t)r->nbuf * sizeof(pw_unit_out) * (size_t)r->n_units + 64);
        const size_t b_tmpl = in->template_atoms > 0 ? up256(template_groups_bytes(nmax)) : 0;
        {
            int rcb = block_take(c, b_off + b_xyz + 2 * b_con + b_tmpl + b_out, &r->block, &r->block_bytes);
            if (rcb != PW_OK) { pw_resident_free(c, r); return rcb; }
        }
        unsigned char* base = (unsigned char*)r->block;
        r->d_offset = (long*)base; base += b_off;
        r->d_xyz = (double*)base; base += b_xyz;
        r->d_vdw = (double*)base; base += b_con;
        r->d_mass = (double*)base; base += b_con;
        std::vector<unsigned char> tmpl_host;
        if (b_tmpl) {
            r->d_tmpl = base; base += b_tmpl;
            tmpl_host.resize(template_groups_bytes(nmax));
            template_groups_build(in->vdw, in->mass, nmax, tmpl_host.data());
        }
        r->d_outs[0] = (pw_unit_out*)base;
        for (int k = 1; k < r->nbuf; ++k) r->d_outs[k] = r->d_outs[0] + (size_t)k * r->n_units;
        r->d_xw_count = (unsigned*)(r->d_outs[0] + (size_t)r->nbuf * r->n_units);   // (extra-window counters)
        for (int k = 0; k < PW_SETS; ++k) r->written_set[k] = -1;
        r->d_out = r->d_outs[0];
        r->cur = 0;
        UP_TRY(hipMemcpyAsync(r->d_offset, in->atom_offset, sizeof(long) * (r->n_units + 1),
                              hipMemcpyHostToDevice, c->stream));
        UP_TRY(hipMemcpyAsync(r->d_xyz, in->xyz, sizeof(double) * 3 * natoms, hipMemcpyHostToDevice,
                              c->stream));
        UP_TRY(hipMemcpyAsync(r->d_vdw, in->vdw, sizeof(double) * nconst, hipMemcpyHostToDevice,
                              c->stream));
        UP_TRY(hipMemcpyAsync(r->d_mass, in->mass, sizeof(double) * nconst, hipMemcpyHostToDevice,
                              c->stream));
        if (r->d_tmpl) UP_TRY(hipMemcpyAsync(r->d_tmpl, tmpl_host.data(), tmpl_host.size(), hipMemcpyHostToDevice, c->stream));
        UP_TRY(hipMemsetAsync(r->d_outs[0], 0, r->nbuf * sizeof(pw_unit_out) * r->n_units + 64, c->stream));
        UP_TRY(hipStreamSynchronize(c->stream));
    }
#undef UP_TRY
    c->need_fork = 1;
    *out = r;
    return PW_OK;
}

// ---- a batch whose coordinates arrive while it is being analysed --------------------------------------------
// One molecule type (template_atoms per unit), n_units known, coordinates appended in unit order.  The analysis
// may be launched right after pw_resident_stream_begin: its chains and average-diameter teams take units in
// index order and wait for the `ready` counter, which every append raises behind its copy on the API stream.
// What this hides is the reader: decoding a 1000-frame HISTORY file takes about as long as a third of the
// analysis' own latency (reference: the frame loop of Trajectory._analysis_serial reads and analyses one frame
// after the other, trajectory.py:496-522).
int pw_resident_stream_begin(pw_context* c, int64_t n_units, int64_t template_atoms, const double* vdw, const double* mass,
                             pw_resident** out) {
    if (!c || !out || n_units <= 0 || template_atoms <= 0 || !vdw || !mass) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    *out = nullptr;
    PW_HOST_UNSUPPORTED(c, "streamed batches");
    if (template_atoms > 40000 || n_units * template_atoms > (int64_t)1 << 40) return PW_E_TOO_LARGE;
    {
        const int pcap = wanted_p_cap(c) > c->p_cap ? wanted_p_cap(c) : c->p_cap;
        if (UnitShared::bytes((int)template_atoms, 1, 1, 1, false, pcap) + 64 > 160 * 1024 - 256 - PW_KERNEL_STATIC_LDS) {
            snprintf(g_err, sizeof(g_err), "molecule with %ld atoms: beyond LDS, upload it in one piece", (long)template_atoms);
            return PW_E_TOO_LARGE;
        }
    }
    PW_ON_DEVICE(c->device);
    pw_resident* r = new (std::nothrow) pw_resident();
    if (!r) return PW_E_NOMEM;
    memset((void*)r, 0, sizeof(*r));
    r->n_units = (long)n_units;
    r->n_atoms = (long)(n_units * template_atoms);
    r->nmax = (int)template_atoms;
    r->vstride = 0;
    r->nbuf = c->nsets ? c->nsets : auto_sets(r->n_units);
    auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_off = up256(sizeof(long) * (size_t)(r->n_units + 1)), b_xyz = up256(sizeof(double) * 3 * (size_t)r->n_atoms);
    const size_t b_con = up256(sizeof(double) * (size_t)template_atoms);
    const size_t b_out = up256((size_t)r->nbuf * sizeof(pw_unit_out) * (size_t)r->n_units + 64);
    const size_t b_tmpl = up256(template_groups_bytes((int)template_atoms));
    int rcb = block_take(c, b_off + b_xyz + 2 * b_con + b_tmpl + b_out + 256, &r->block, &r->block_bytes);
    if (rcb != PW_OK) { pw_resident_free(c, r); return rcb; }
    unsigned char* base = (unsigned char*)r->block;
    r->d_offset = (long*)base; base += b_off;
    r->d_xyz = (double*)base; base += b_xyz;
    r->d_vdw = (double*)base; base += b_con;
    r->d_mass = (double*)base; base += b_con;
    r->d_tmpl = base; base += b_tmpl;
    std::vector<unsigned char> tmpl_host(template_groups_bytes((int)template_atoms));
    template_groups_build(vdw, mass, (int)template_atoms, tmpl_host.data());
    r->d_outs[0] = (pw_unit_out*)base; base += b_out;
    {
        // the counter lives in host memory the device can read: the host raises it when a copy has landed, and no
        // kernel or copy has to be scheduled beside the launch that is waiting for it
        constexpr unsigned RING = 4096;
        if (!c->ready_vals) {
            hipError_t eh = hipHostMalloc((void**)&c->ready_vals, RING * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
            if (eh == hipSuccess) eh = hipHostGetDevicePointer((void**)&c->ready_dev, c->ready_vals, 0);
            if (eh != hipSuccess) { set_err("pw_resident_stream_begin (ready counters)", eh); c->ready_vals = nullptr; pw_resident_free(c, r); return PW_E_HIP; }
        }
        const unsigned at = c->ready_at++ % RING;
        r->h_ready = c->ready_vals + at;
        r->d_ready = c->ready_dev + at;
        __atomic_store_n(r->h_ready, 0ull, __ATOMIC_RELEASE);
    }
    for (int k = 1; k < r->nbuf; ++k) r->d_outs[k] = r->d_outs[0] + (size_t)k * r->n_units;
    r->d_xw_count = (unsigned*)(r->d_outs[0] + (size_t)r->nbuf * r->n_units);
    for (int k = 0; k < PW_SETS; ++k) r->written_set[k] = -1;
    r->d_out = r->d_outs[0];
    // atom offsets of a uniform batch: k * atoms (built on the host: 8 KB per 1000 units)
    std::vector<long> off((size_t)r->n_units + 1);
    for (long u = 0; u <= r->n_units; ++u) off[(size_t)u] = u * (long)template_atoms;
    hipError_t e = hipMemcpyAsync(r->d_offset, off.data(), sizeof(long) * (r->n_units + 1), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_vdw, vdw, sizeof(double) * template_atoms, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_mass, mass, sizeof(double) * template_atoms, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_tmpl, tmpl_host.data(), tmpl_host.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_outs[0], 0, r->nbuf * sizeof(pw_unit_out) * r->n_units + 64, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // (`off` is a local; 10 us of copies)
    if (e != hipSuccess) { set_err("pw_resident_stream_begin", e); pw_resident_free(c, r); return PW_E_HIP; }
    c->need_fork = 1;
    *out = r;
    return PW_OK;
}

// coordinates of units [first, first + count): first must be what has been appended so far.  Returns when the copy
// has landed (from page-locked memory -- pw_context_pinned -- a DMA of tens of microseconds per megabyte).
int pw_resident_stream_append(pw_context* c, pw_resident* r, const double* xyz, int64_t first, int64_t count) {
    if (!c || !r || !xyz || count < 0) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (!r->d_ready || first != r->ready_units || first + count > r->n_units) {
        snprintf(g_err, sizeof(g_err), "pw_resident_stream_append: units %ld..%ld do not continue the %ld appended so far (of %ld)",
                 (long)first, (long)(first + count), r->ready_units, r->n_units);
        return PW_E_BAD_ARG;
    }
    if (count == 0) return PW_OK;
    PW_ON_DEVICE(c->device);
    const size_t per = (size_t)r->nmax * 3;
    HIP_TRY(hipMemcpyAsync(r->d_xyz + (size_t)first * per, xyz, sizeof(double) * per * (size_t)count, hipMemcpyHostToDevice, c->stream));
    // the copy has landed when the API stream is idle again (nothing else is queued on it: the launches run on the
    // pipeline's own streams); then the counter is raised from the host
    HIP_TRY(hipStreamSynchronize(c->stream));
    r->ready_units = (long)(first + count);
    __atomic_store_n(r->h_ready, (unsigned long long)r->ready_units, __ATOMIC_RELEASE);
    if (r->pending && r->ready_units == r->n_units) return pw_resident_launch(c, r, r->pending_stages);   // (see there)
    return PW_OK;
}

// a batch whose arrays are already on the device (pw_resident_from_cells); takes ownership of the four
// blocks (taken from the context's cache: part_bytes says how large each is)
int pw_internal_resident_adopt(pw_context* c, long n_units, long n_atoms, int nmax, long* d_offset, double* d_xyz,
                               double* d_vdw, double* d_mass, const size_t* part_bytes, pw_resident** out) {
    if (!c || !out || n_units <= 0 || nmax <= 0 || !part_bytes) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "device batches");
    PW_ON_DEVICE(c->device);
    pw_resident* r = new (std::nothrow) pw_resident();
    if (!r) return PW_E_NOMEM;
    memset((void*)r, 0, sizeof(*r));
    r->nbuf = c->nsets ? c->nsets : auto_sets(n_units);
    void* outs = nullptr;
    size_t outs_bytes = 0;
    int rcb = block_take(c, r->nbuf * sizeof(pw_unit_out) * (size_t)n_units + 64, &outs, &outs_bytes);
    if (rcb != PW_OK) { delete r; return rcb; }
    r->d_outs[0] = (pw_unit_out*)outs;
    hipError_t e = hipMemsetAsync(r->d_outs[0], 0, r->nbuf * sizeof(pw_unit_out) * n_units + 64, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        set_err("pw_internal_resident_adopt", e);
        block_give(c, outs, outs_bytes);
        delete r;
        return PW_E_HIP;
    }
    r->n_units = n_units; r->n_atoms = n_atoms; r->nmax = nmax; r->vstride = 1;
    r->d_offset = d_offset; r->d_xyz = d_xyz; r->d_vdw = d_vdw; r->d_mass = d_mass;
    r->parts[0] = d_offset; r->parts[1] = d_xyz; r->parts[2] = d_vdw; r->parts[3] = d_mass; r->parts[4] = outs;
    for (int k = 0; k < 4; ++k) r->part_bytes[k] = part_bytes[k];
    r->part_bytes[4] = outs_bytes;
    for (int k = 1; k < r->nbuf; ++k) r->d_outs[k] = r->d_outs[0] + (size_t)k * n_units;
    r->d_xw_count = (unsigned*)(r->d_outs[0] + (size_t)r->nbuf * n_units);
    for (int k = 0; k < PW_SETS; ++k) r->written_set[k] = -1;
    r->d_out = r->d_outs[0];
    r->cur = 0;
    c->need_fork = 1;
    *out = r;
    return PW_OK;
}

// Windows beyond the PW_W_MAX a record holds (pw_unit.hpp: stage_windows appends them to the launch's list).
// The list's device buffer only exists once a launch has asked for it: the first launch that meets such a
// unit counts them, the download grows the buffer and reports PW_E_RETRY, the repeated launch fills it.
static int fetch_extra_windows(pw_context* c, pw_resident* r, unsigned count) {
    c->extra->clear();
    if (count == 0) return PW_OK;
    if (count > r->xw_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (r->d_xw[0]) HIP_TRY(hipFree(r->d_xw[0]));
        for (int k = 0; k < PW_SETS; ++k) r->d_xw[k] = nullptr;
        r->xw_cap = 0;
        const unsigned cap = ((count + count / 2 + 1023u) / 1024u) * 1024u;
        HIP_TRY(hipMalloc((void**)&r->d_xw[0], (size_t)r->nbuf * cap * sizeof(pw_extra_window)));
        for (int k = 1; k < r->nbuf; ++k) r->d_xw[k] = r->d_xw[0] + (size_t)k * cap;
        r->xw_cap = cap;
        snprintf(g_err, sizeof(g_err), "%u windows beyond the %d a record holds: launch the analysis again "
                 "(the list for them has been allocated)", count, PW_W_MAX);
        return PW_E_RETRY;
    }
    c->extra->resize(count);
    HIP_TRY(hipMemcpy(c->extra->data(), r->d_xw[r->cur], (size_t)count * sizeof(pw_extra_window), hipMemcpyDeviceToHost));
    std::sort(c->extra->begin(), c->extra->end(), [](const pw_extra_window& a, const pw_extra_window& b) {
        return a.unit != b.unit ? a.unit < b.unit : a.index < b.index;
    });
    return PW_OK;
}

int pw_resident_download(pw_context* c, pw_resident* r, pw_unit_out* out) {
    if (!c || !r || !out) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (r->n_units == 0) return PW_OK;
    if (c->device < 0) {
        if (!r->host) return PW_E_BAD_ARG;
        memcpy(out, r->host->out.data(), sizeof(pw_unit_out) * (size_t)r->n_units);
        return PW_OK;
    }
    if (r->d_ready && r->ready_units < r->n_units) {
        snprintf(g_err, sizeof(g_err), "streamed batch: %ld of %ld units appended", r->ready_units, r->n_units);
        return PW_E_BAD_ARG;
    }
    PW_ON_DEVICE(c->device);
    // wait for the launch that wrote these records -- not for launches of other batches issued since
    // (a trajectory analysed in pieces downloads piece k while piece k + 1 is still running)
    const int ws = r->written_set[r->cur];
    if (ws >= 0 && c->done_valid[ws]) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done[ws], 0));
    } else {
        int rcj = join_pipeline(c);
        if (rcj != PW_OK) return rcj;
    }
    unsigned xw_count = 0;
    HIP_TRY(hipMemcpyAsync(out, r->d_out, sizeof(pw_unit_out) * r->n_units, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipMemcpyAsync(&xw_count, r->d_xw_count + r->cur, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = check_queue_error(c);
    if (rc != PW_OK) return rc;
    return fetch_extra_windows(c, r, xw_count);
}

// The same hand-over without the records: for callers that read the records on the device (the RCCL
// gather).  Waits for the latest launch of the batch, reports a timed-out window launch (the only other
// place that is checked is the download), and loads the launch's extra windows into the context's list.
int pw_resident_extra_windows(pw_context* c, pw_resident* r, int64_t* count) {
    if (!c || !r) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (count) *count = 0;
    if (r->n_units == 0) return PW_OK;
    if (c->device < 0) { if (count) *count = (int64_t)c->extra->size(); return PW_OK; }
    PW_ON_DEVICE(c->device);
    const int ws = r->written_set[r->cur];
    if (ws >= 0 && c->done_valid[ws]) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_done[ws], 0));
    } else {
        int rcj = join_pipeline(c);
        if (rcj != PW_OK) return rcj;
    }
    unsigned xw_count = 0;
    HIP_TRY(hipMemcpyAsync(&xw_count, r->d_xw_count + r->cur, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = check_queue_error(c);
    if (rc != PW_OK) return rc;
    rc = fetch_extra_windows(c, r, xw_count);
    if (rc == PW_OK && count) *count = (int64_t)c->extra->size();
    return rc;
}

int64_t pw_context_extra_windows(pw_context* c, pw_extra_window* buf, int64_t cap) {
    if (!c || !c->extra) return 0;
    PW_LOCK_CONTEXT(c);
    const int64_t n = (int64_t)c->extra->size();
    if (buf && cap > 0) memcpy(buf, c->extra->data(), (size_t)(n < cap ? n : cap) * sizeof(pw_extra_window));
    return n;
}

/* 1: analyses on this context run as the overlapped three-launch pipeline; 0: as single launches (PW_FUSED=1,
 * or the probe of pw_context_create found fewer concurrent streams than the pipeline needs) */
int pw_context_pipelined(pw_context* c) { return c && !c->fused ? 1 : 0; }

int pw_context_point_capacity(pw_context* c) { return c ? (c->p_cap > 0 ? c->p_cap : wanted_p_cap(c)) : 0; }

int pw_context_reserve_points(pw_context* c, int64_t n_points) {
    if (!c || n_points < 0 || n_points > 4000000) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    const int want = round_p_cap((long)n_points);
    if (want > c->p_cap_min) c->p_cap_min = want;
    return PW_OK;
}

void pw_resident_free(pw_context* c, pw_resident* r) {
    if (!r) return;
    PW_LOCK_CONTEXT(c);
    if (r->host) { delete r->host; delete r; return; }
    DeviceScope scope;
    if (c) (void)scope.enter(c->device);
    // a streamed batch given up before its last unit: the launches that wait for units stop waiting (the top bit of
    // the counter; they would give up by themselves after PW_STREAM_LIMIT_MS)
    const bool given_up = r->h_ready && r->ready_units < r->n_units;
    if (given_up) __atomic_store_n(r->h_ready, (1ull << 63) | (unsigned long long)r->ready_units, __ATOMIC_RELEASE);
    // wait for the launches that touched this batch -- not for the whole device -- and keep its blocks for
    // the next one
    if (c) {
        for (int k = 0; k < PW_SETS; ++k) {
            const int ws = r->written_set[k];
            if (ws >= 0 && c->done_valid[ws]) (void)hipEventSynchronize(c->ev_done[ws]);
            if (r->read_valid[k]) (void)hipEventSynchronize(r->ev_read[k]);
            // (what those launches flagged when they were told to stop is not an error of anybody's next analysis)
            if (given_up && ws >= 0 && c->queue) {
                (void)hipMemsetAsync(&c->queue[ws].error, 0, sizeof(c->queue[ws].error), c->stream);
            }
        }
        (void)hipStreamSynchronize(c->stream);
    }
    if (r->block) block_give(c, r->block, r->block_bytes);
    for (int k = 0; k < 5; ++k)
        if (r->parts[k]) block_give(c, r->parts[k], r->part_bytes[k]);
    if (r->d_xw[0]) (void)hipFree(r->d_xw[0]);
    for (int k = 0; k < PW_SETS; ++k)
        if (r->ev_read[k]) (void)hipEventDestroy(r->ev_read[k]);
    delete r;
}

// How many of the pipeline's pacing gates gave up waiting since the context was created (tail | head << 16 |
// residency << 32).  A gate only paces launches, it is never a dependency: a time-out costs 20 ms and nothing else.
// Diagnostic: zero on a healthy device.
int pw_context_gate_timeouts(pw_context* c, uint64_t* count) {
    if (!c || !count) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    *count = 0;
    if (c->device < 0 || !c->counter) return PW_OK;
    PW_ON_DEVICE(c->device);
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, c->counter + 4 * PW_SETS + 2, sizeof(v), hipMemcpyDeviceToHost));
    *count = (uint64_t)v;
    return PW_OK;
}

// for the translation units that have no error text of their own (pw_history.cpp): the calling thread's
extern "C" void pw_internal_set_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }
static void count_retry(pw_context* c) {
    c->retries += 1;
    g_retries_total.fetch_add(1);
    fprintf(stderr, "pywindow_amd: analysis repeated after a time-out: %s\n", g_err);
}
int pw_context_retries(pw_context* c, uint64_t* count) {
    if (!c || !count) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    *count = (uint64_t)c->retries;
    return PW_OK;
}
int pw_context_count_retry(pw_context* c) {
    if (!c) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    count_retry(c);
    return PW_OK;
}
uint64_t pw_retries_total(void) { return (uint64_t)g_retries_total.load(); }
int pw_context_queue_state(pw_context* c, uint64_t* out, int cap) {
    if (!c || !out || cap < 4 * PW_SETS) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    for (int k = 0; k < 4 * PW_SETS; ++k) out[k] = 0;
    if (c->device < 0 || !c->queue) return PW_OK;
    PW_ON_DEVICE(c->device);
    UnitQueue q[PW_SETS];
    HIP_TRY(hipMemcpy(q, c->queue, sizeof(q), hipMemcpyDeviceToHost));
    for (int b = 0; b < PW_SETS; ++b) {
        out[4 * b] = q[b].head; out[4 * b + 1] = q[b].tail; out[4 * b + 2] = (uint64_t)q[b].started; out[4 * b + 3] = (uint64_t)q[b].error;
    }
    return PW_OK;
}

void* pw_resident_device_results(pw_resident* r) { return r ? (r->host ? (void*)r->host->out.data() : (void*)r->d_out) : nullptr; }
int64_t pw_resident_units(pw_resident* r) { return r ? r->n_units : 0; }

// Stream-ordered hand-over of the latest results to a caller's stream (the RCCL gather of a
// one-process-per-GPU job runs on PyTorch's stream): no host synchronisation on either side.
int pw_resident_results_ready(pw_context* c, pw_resident* r, void* stream, void** results) {
    if (!c || !r) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "stream-ordered hand-over");
    PW_ON_DEVICE(c->device);
    hipStream_t ext = (hipStream_t)stream;
    for (int b = 0; b < PW_SETS; ++b)
        if (c->done_valid[b]) HIP_TRY(hipStreamWaitEvent(ext, c->ev_done[b], 0));
    // single-launch analyses, uploads and downloads go through the API stream
    HIP_TRY(hipEventRecord(c->ev_ext, c->stream));
    HIP_TRY(hipStreamWaitEvent(ext, c->ev_ext, 0));
    if (results) *results = (void*)r->d_out;
    return PW_OK;
}

int pw_resident_results_release(pw_context* c, pw_resident* r, void* stream) {
    if (!c || !r) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "stream-ordered hand-over");
    PW_ON_DEVICE(c->device);
    // the launch that next writes this result buffer (two launches of this batch from now) waits for
    // what the caller's stream has been given so far; launches in between are not held back
    const int k = r->cur;
    if (!r->ev_read[k]) HIP_TRY(hipEventCreateWithFlags(&r->ev_read[k], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->ev_read[k], (hipStream_t)stream));
    r->read_valid[k] = 1;
    return PW_OK;
}

int pw_resident_time(pw_context* c, pw_resident* r, uint32_t stages, int iters, float* ms) {
    if (!c || !r || !ms || iters < 1) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (c->device < 0) {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < iters; ++i) {
            int rch = pw_resident_launch(c, r, stages);
            if (rch != PW_OK) return rch;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        *ms = (float)(((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / iters);
        return PW_OK;
    }
    PW_ON_DEVICE(c->device);
    int rc = pw_resident_launch(c, r, stages);  // warm-up, also sizes the workspace
    if (rc != PW_OK) return rc;
    rc = pw_resident_sync(c);
    if (rc != PW_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    c->need_fork = 1;        // the first timed launch starts after ev0 on every stream
    for (int i = 0; i < iters; ++i) {
        rc = pw_resident_launch(c, r, stages);
        if (rc != PW_OK) return rc;
    }
    rc = join_pipeline(c);
    if (rc != PW_OK) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float total = 0.f;
    HIP_TRY(hipEventElapsedTime(&total, c->ev0, c->ev1));
    *ms = total / (float)iters;
    return check_queue_error(c);     // a timed-out launch must not be reported as a time
}

// One analysis on its own (nothing else in flight), HIP events on the stream of each of its three
// launches: ms[0] optimiser chains, ms[1] average diameter, ms[2] window search (a consumer: it runs
// from the moment the chains are resident until the last published unit is fitted).
int pw_resident_stage_times(pw_context* c, pw_resident* r, float* ms) {
    if (!c || !r || !ms) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "per-launch timing");
    if (c->fused) { snprintf(g_err, sizeof(g_err), "PW_FUSED=1: the analysis is one launch"); return PW_E_BAD_ARG; }
    PW_ON_DEVICE(c->device);
    for (int k = 0; k < 3; ++k)
        for (int e = 0; e < 2; ++e)
            if (!c->ev_t[k][e]) HIP_TRY(hipEventCreate(&c->ev_t[k][e]));
    int rc = pw_resident_launch(c, r, PW_STAGE_ALL);     // warm-up, sizes the workspaces
    if (rc == PW_OK) rc = pw_resident_sync(c);
    if (rc != PW_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
    c->timing = 1;
    rc = pw_resident_launch(c, r, PW_STAGE_ALL);
    c->timing = 0;
    if (rc == PW_OK) rc = pw_resident_sync(c);
    if (rc != PW_OK) return rc;
    for (int k = 0; k < 3; ++k) {
        if (k == 1 && !c->timed_avg) { ms[k] = 0.f; continue; }      // (no launch of its own: a stage of the window teams)
        HIP_TRY(hipEventElapsedTime(&ms[k], c->ev_t[k][0], c->ev_t[k][1]));
    }
    return PW_OK;
}

// launch + download with the two capacities that can only be known afterwards: a unit that wants more
// sampling vectors than the adjust knobs imply (a sphere of several thousand angstroms) raises the
// context's minimum, a unit with more windows than a record holds gets the launch's list allocated --
// either way the launch is repeated, so no limit of the engine ever shows in a result
static int launch_and_download(pw_context* c, pw_resident* r, uint32_t stages, pw_unit_out* out) {
    int rc = PW_OK;
    int timeouts = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        rc = pw_resident_launch(c, r, stages);
        if (rc == PW_OK) rc = pw_resident_download(c, r, out);
        if (rc == PW_E_RETRY) continue;
        // (a launch that saw another launch of the same analysis make no progress for a whole limit: the analysis is
        // repeated, up to PW_TIMEOUT_REPEATS times, every repeat counted)
        if (rc == PW_E_TIMEOUT && timeouts < c->timeout_repeats) { ++timeouts; --attempt; count_retry(c); continue; }
        if (rc != PW_OK) return rc;
        long want = 0;
        for (long u = 0; u < r->n_units; ++u)
            if (out[u].status & PW_ST_POINTS_OVERFLOW) {
                if (out[u].n_points > want) want = out[u].n_points;
                if (out[u].n_points_avg > want) want = out[u].n_points_avg;
            }
        if (want <= c->p_cap) return PW_OK;      // (nothing flagged, or flagged for a reason a larger workspace does not cure)
        c->p_cap_min = round_p_cap(want);
    }
    return rc;
}

int pw_analysis_batch(pw_context* c, const pw_batch_in* in, uint32_t stages, pw_unit_out* out) {
    if (!c || !in || !out) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (c->device < 0) return host_analyse(c, in, stages, out, nullptr);
    pw_resident* r = nullptr;
    int rc = pw_resident_upload(c, in, &r);
    if (rc != PW_OK) return rc;
    rc = launch_and_download(c, r, stages, out);
    pw_resident_free(c, r);
    return rc;
}

int pw_analysis_debug(pw_context* c, const pw_batch_in* in, uint32_t stages, pw_unit_out* out, pw_unit_debug* dbg) {
    if (!c || !in || !out || !dbg) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (in->n_units == 0) return PW_OK;
    if (c->device < 0) {
        memset(dbg, 0, sizeof(pw_unit_debug) * (size_t)in->n_units);
        const int keep = c->host_threads;
        c->host_threads = 1;                     // (test instrumentation: one unit at a time)
        int rch = host_analyse(c, in, stages, out, dbg);
        c->host_threads = keep;
        return rch;
    }
    PW_ON_DEVICE(c->device);
    pw_resident* r = nullptr;
    int rc = pw_resident_upload(c, in, &r);
    if (rc != PW_OK) return rc;
    pw_unit_debug* d_dbg = nullptr;
    const size_t bytes = sizeof(pw_unit_debug) * (size_t)in->n_units;
    auto set_debug = [&](pw_unit_debug* p) -> int {
        // (after a sync: no team is running while the pointers change)
        HIP_TRY(hipDeviceSynchronize());
        if (c->ws_blocks > 0) {
            hipLaunchKernelGGL(pw_set_debug_kernel, dim3((c->ws_blocks + 255) / 256), dim3(256), 0, c->stream, c->ws,
                               c->ws_blocks, p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        return PW_OK;
    };
    hipError_t e = hipMalloc((void**)&d_dbg, bytes);
    if (e == hipSuccess) e = hipMemset(d_dbg, 0, bytes);
    if (e != hipSuccess) { set_err("pw_analysis_debug", e); pw_resident_free(c, r); return PW_E_HIP; }
    // a first launch sizes the workspaces (they are re-allocated, zeroed, when they grow); the capture
    // run follows with every workspace pointing at the buffer
    rc = pw_resident_launch(c, r, stages);
    if (rc == PW_OK) rc = pw_resident_sync(c);
    if (rc == PW_OK) rc = launch_and_download(c, r, stages, out);      // (settles the capacities)
    if (rc == PW_OK) rc = set_debug(d_dbg);
    if (rc == PW_OK) rc = pw_resident_launch(c, r, stages);
    if (rc == PW_OK) rc = pw_resident_download(c, r, out);
    int rc2 = set_debug(nullptr);
    if (rc == PW_OK) rc = rc2;
    if (rc == PW_OK) {
        e = hipMemcpy(dbg, d_dbg, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_err("pw_analysis_debug download", e); rc = PW_E_HIP; }
    }
    (void)hipFree(d_dbg);
    pw_resident_free(c, r);
    return rc;
}

int pw_point_gaps(pw_context* c, const pw_batch_in* in, const int64_t* unit_of_point,
                  const double* points, int64_t n_points, double* gap, int32_t* argmin) {
    if (!c || !in || !unit_of_point || !points || !gap || !argmin || n_points < 0) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    if (n_points == 0) return PW_OK;
    for (int64_t q = 0; q < n_points; ++q)
        if (unit_of_point[q] < 0 || unit_of_point[q] >= in->n_units) return PW_E_BAD_ARG;
    if (c->device < 0) {
        const int vs = in->template_atoms > 0 ? 0 : 1;
        for (int64_t q = 0; q < n_points; ++q) {
            const int64_t a0 = in->atom_offset[unit_of_point[q]];
            const int n = (int)(in->atom_offset[unit_of_point[q] + 1] - a0);
            const double px = points[3 * q], py = points[3 * q + 1], pz = points[3 * q + 2], pp = sq3(px, py, pz);
            double best = PW_INF;
            int bi = 0;
            for (int i = 0; i < n; ++i) {
                const double x = in->xyz[3 * (a0 + i)], y = in->xyz[3 * (a0 + i) + 1], z = in->xyz[3 * (a0 + i) + 2];
                const double g = pw_fma(z, pz, pw_fma(x, px, y * py));
                const double d2 = pw_m2add(g, sq3(x, y, z)) + pp;
                const double v = pw_sqrt(d2 > 0.0 ? d2 : 0.0) - in->vdw[a0 * vs + i];
                if (v < best) { best = v; bi = i; }
            }
            gap[q] = best;
            argmin[q] = bi;
        }
        return PW_OK;
    }
    PW_ON_DEVICE(c->device);
    pw_resident* r = nullptr;
    int rc = pw_resident_upload(c, in, &r);
    if (rc != PW_OK) return rc;
    long* d_u = nullptr;
    double *d_p = nullptr, *d_g = nullptr;
    int* d_a = nullptr;
    auto cleanup = [&]() {
        if (d_u) (void)hipFree(d_u);
        if (d_p) (void)hipFree(d_p);
        if (d_g) (void)hipFree(d_g);
        if (d_a) (void)hipFree(d_a);
        pw_resident_free(c, r);
    };
#define PG_TRY(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            set_err(#call, e_);                        \
            cleanup();                                 \
            return PW_E_HIP;                           \
        }                                              \
    } while (0)
    PG_TRY(hipMalloc((void**)&d_u, sizeof(long) * n_points));
    PG_TRY(hipMalloc((void**)&d_p, sizeof(double) * 3 * n_points));
    PG_TRY(hipMalloc((void**)&d_g, sizeof(double) * n_points));
    PG_TRY(hipMalloc((void**)&d_a, sizeof(int) * n_points));
    PG_TRY(hipMemcpyAsync(d_u, unit_of_point, sizeof(long) * n_points, hipMemcpyHostToDevice, c->stream));
    PG_TRY(hipMemcpyAsync(d_p, points, sizeof(double) * 3 * n_points, hipMemcpyHostToDevice, c->stream));
    int block = 256;
    long grid = (n_points + block - 1) / block;
    hipLaunchKernelGGL(pw_point_gap_kernel, dim3((unsigned)grid), dim3(block), 0, c->stream,
                       (long)n_points, d_u, d_p, r->d_offset, r->d_xyz, r->d_vdw, d_g, d_a, r->vstride);
    PG_TRY(hipGetLastError());
    PG_TRY(hipMemcpyAsync(gap, d_g, sizeof(double) * n_points, hipMemcpyDeviceToHost, c->stream));
    PG_TRY(hipMemcpyAsync(argmin, d_a, sizeof(int) * n_points, hipMemcpyDeviceToHost, c->stream));
    PG_TRY(hipStreamSynchronize(c->stream));
#undef PG_TRY
    cleanup();
    return PW_OK;
}

int pw_dbscan(pw_context* c, const double* points, int64_t n, double eps, int mode, int32_t* labels,
              int32_t* n_clusters) {
    if (!c || !labels || !n_clusters || n < 0 || n > PW_DBSCAN_MAX || (n > 0 && !points)) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    *n_clusters = 0;
    if (n == 0) return PW_OK;
    PW_HOST_UNSUPPORTED(c, "pw_dbscan");
    PW_ON_DEVICE(c->device);
    double* d_p = nullptr;
    int *d_i = nullptr, *d_l = nullptr;
    TeamWorkspace* d_ws = nullptr;
    unsigned long long* d_adj = nullptr;
    auto cleanup = [&]() {
        if (d_p) (void)hipFree(d_p);
        if (d_i) (void)hipFree(d_i);
        if (d_l) (void)hipFree(d_l);
        if (d_ws) (void)hipFree(d_ws);
        if (d_adj) (void)hipFree(d_adj);
    };
#define DB_TRY(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            set_err(#call, e_);                        \
            cleanup();                                 \
            return PW_E_HIP;                           \
        }                                              \
    } while (0)
    std::vector<double> soa((size_t)3 * n);
    std::vector<int> ident((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) soa[(size_t)k * n + i] = points[3 * i + k];
        ident[i] = (int)i;
    }
    DB_TRY(hipMalloc((void**)&d_p, sizeof(double) * 3 * n));
    DB_TRY(hipMalloc((void**)&d_i, sizeof(int) * n));
    DB_TRY(hipMalloc((void**)&d_l, sizeof(int) * (n + 1)));
    DB_TRY(hipMalloc((void**)&d_ws, sizeof(TeamWorkspace)));
    const int p_cap = round_p_cap((long)n);
    DB_TRY(hipMalloc((void**)&d_adj, sizeof(unsigned long long) * team_adj_words(p_cap)));
    DB_TRY(hipMemcpyAsync(d_p, soa.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    DB_TRY(hipMemcpyAsync(d_i, ident.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    const int lds_bytes = (mode & 2) ? 0 : 96 * 1024;
    if (mode & 1) {
        DB_TRY(hipFuncSetAttribute((const void*)pw_dbscan_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        hipLaunchKernelGGL(pw_dbscan_kernel<1>, dim3(1), dim3(64), lds_bytes, c->stream, d_p, (int)n, eps, lds_bytes, d_ws, p_cap,
                           d_adj, d_i, d_l, d_l + n);
    } else {
        DB_TRY(hipFuncSetAttribute((const void*)pw_dbscan_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        hipLaunchKernelGGL(pw_dbscan_kernel<4>, dim3(1), dim3(256), lds_bytes, c->stream, d_p, (int)n, eps, lds_bytes, d_ws, p_cap,
                           d_adj, d_i, d_l, d_l + n);
    }
    DB_TRY(hipGetLastError());
    DB_TRY(hipMemcpyAsync(labels, d_l, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    DB_TRY(hipMemcpyAsync(n_clusters, d_l + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    DB_TRY(hipStreamSynchronize(c->stream));
#undef DB_TRY
    cleanup();
    return PW_OK;
}

int pw_pairwise_sum(pw_context* c, const double* values, int64_t n, int mode, double* sum) {
    if (!c || !sum || n < 0 || n > 0x7fffffff || (n > 0 && !values)) return PW_E_BAD_ARG;
    PW_LOCK_CONTEXT(c);
    PW_HOST_UNSUPPORTED(c, "pw_pairwise_sum");
    PW_ON_DEVICE(c->device);
    double *d_a = nullptr, *d_s = nullptr;
    auto cleanup = [&]() {
        if (d_a) (void)hipFree(d_a);
        if (d_s) (void)hipFree(d_s);
    };
#define PS_TRY(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            set_err(#call, e_);                        \
            cleanup();                                 \
            return PW_E_HIP;                           \
        }                                              \
    } while (0)
    const size_t scratch_doubles = 8 * 160 + 256 + 324 / 2 + 2;
    PS_TRY(hipMalloc((void**)&d_a, sizeof(double) * (size_t)(n > 0 ? n : 1)));
    PS_TRY(hipMalloc((void**)&d_s, sizeof(double) * scratch_doubles));
    if (n > 0) PS_TRY(hipMemcpyAsync(d_a, values, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (mode & 1)
        hipLaunchKernelGGL(pw_pairwise_sum_kernel<1>, dim3(1), dim3(64), 0, c->stream, d_a, (long)n, d_s + 1, (mode >> 1) & 1, d_s);
    else
        hipLaunchKernelGGL(pw_pairwise_sum_kernel<4>, dim3(1), dim3(256), 0, c->stream, d_a, (long)n, d_s + 1, (mode >> 1) & 1, d_s);
    PS_TRY(hipGetLastError());
    PS_TRY(hipMemcpyAsync(sum, d_s, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PS_TRY(hipStreamSynchronize(c->stream));
#undef PS_TRY
    cleanup();
    return PW_OK;
}

/* diagnostic builds (-DPW_PROFILE): sum of the in-kernel stage timers over all teams, in
 * 100 MHz ticks; zeroes them afterwards.  Returns PW_E_BAD_ARG in normal builds. */
int pw_debug_stage_ticks(pw_context* c, unsigned long long* out32) {
#ifdef PW_PROFILE
    if (!c || !out32 || !c->ws) return PW_E_BAD_ARG;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 32; ++i) out32[i] = 0;
    for (int b = 0; b < c->ws_blocks; ++b) {
        unsigned long long tmp[32];
        HIP_TRY(hipMemcpy(tmp, c->ws[b].prof, sizeof(tmp), hipMemcpyDeviceToHost));
        for (int i = 0; i < 32; ++i) out32[i] += tmp[i];
        HIP_TRY(hipMemset(c->ws[b].prof, 0, sizeof(tmp)));
    }
    return PW_OK;
#else
    (void)c; (void)out32;
    return PW_E_BAD_ARG;
#endif
}

}  // extern "C"
