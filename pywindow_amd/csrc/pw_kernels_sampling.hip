// pw_kernels_sampling.hip -- the sampling launch of the split window search.
//
// find_windows (utilities.py:1364-1553) has two halves of different shape.  Up to the clustering it is bulk
// loops -- the sampling sphere, the DBSCAN radius, ray tests, path scans, DBSCAN (:1374-1487, :1221): lanes over
// sampling vectors, atoms broadcast from LDS, no optimiser state.  The fits of the clusters (:1191-1361) are
// serial optimiser runs, one wave each.  Compiled together, the register budget of the whole was set by the
// optimisers (256 VGPRs, two waves per SIMD).  This translation unit holds ONLY the first half -- the same
// source (pw_unit.hpp: windows_bulk_impl) in a namespace of its own, so that every function it calls is compiled
// for THIS kernel's budget: three waves per SIMD (PW_OCC_S), no optimiser code at all.  The teams take units in
// the order the optimiser chains publish them (UnitQueue), write each unit's clusters into its FitTicket and
// publish one item per cluster to the fit workers (FitQueue; pw_kernels.hip: pw_worker_kernel).
#define pw pw_smp          // a namespace of its own: its inline functions must not merge with pw_kernels.hip's
#ifndef PW_NO_TEAM_STATE_IN_LDS
#define PW_TEAM_STATE_IN_LDS 1
#endif
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <atomic>

#include "../../include/pywindow_amd.h"
#include "pw_unit.hpp"
#include "pw_launch.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip (512 bytes, thread local)

#ifndef PW_OCC_S
#define PW_OCC_S 3
#endif

namespace {

// a value every lane of the wave holds alike, as the scalar it is (branches on it are uniform branches)
__device__ inline long wave_uniform(long v) {
    union { long l; int i[2]; } a;
    a.l = v;
    a.i[0] = __builtin_amdgcn_readfirstlane(a.i[0]);
    a.i[1] = __builtin_amdgcn_readfirstlane(a.i[1]);
    return a.l;
}

constexpr unsigned MASK_SAMPLING = PW_STAGE_WIN_BULK | PW_STAGE_REUSE_OPT | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;

__global__ void __launch_bounds__(256, PW_OCC_S)
pw_sampling_kernel(long n_units, const long* __restrict__ atom_offset, const double* __restrict__ xyz,
                   const double* __restrict__ vdw, const double* __restrict__ mass, int nmax, int nrot, int nlb,
                   PwWsArgs wsa, pw_unit_out* __restrict__ out, UnitQueue* queue, int* __restrict__ slots, pw_params prm_in,
                   const unsigned* __restrict__ rsq_tab, int vstride, FitArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ long s_unit;
    // the team's table of pointers and its parameters live in LDS (PW_TEAM_STATE_IN_LDS): nothing of the kernel's
    // own is on a stack
    __shared__ UnitShared s_sh;
    __shared__ pw_params s_prm;
    using T = DeviceTeam<4>;
    TeamWorkspace* ws = (TeamWorkspace*)wsa.ws + blockIdx.x;
    if (threadIdx.x == 0) {
        // one frame, shifted in place; the DBSCAN bit sets but no per-cluster arrays (they are the ticket's); the
        // window frames and optimiser blocks a fused team would carve are the scratch arena of the sampling stages
        s_sh.carve(lds, nmax, nrot, nlb, 1, 2, wsa.p_cap);
        s_prm = prm_in;
        bind_workspace(ws, wsa, blockIdx.x, rsq_tab, team_slab_bytes(wsa.p_cap), team_adj_words(wsa.p_cap));
    }
    __syncthreads();
    UnitShared& sh = s_sh;
    const pw_params& prm = s_prm;
    // Thread 0 is the team's only contact with the queues: it takes a unit at the top of an iteration and publishes
    // at the end, and a TEAM BARRIER separates the two regions.  Without it the two thread-0 regions sit either side
    // of the loop's back edge and the compiler threads lane 0 straight from one into the other: the wave then reaches
    // the barrier below without it (seen on gfx950 / ROCm 7.2 -- lane 0 masked off from a team's second unit on, its
    // registers stale; and with the fetch moved to the END of the iteration, intermittent device faults as soon as
    // launches overlapped).  Fetch at the top, publish at the end, barrier after both: the shape of the window
    // launch of rounds 1-3.
    auto take_unit = [&]() {
        // consumer of the optimiser launch: ONE relaxed poll loop -> agent acquire -> (team barrier) -> loads
        long pos = (long)atomicAdd(&queue->head, 1ull);
        long u = -1;
        if (pos < n_units) {
            long long t0 = wall_clock64();
            for (;;) {
                int v = __hip_atomic_load(&slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= 0) { u = v; break; }
                __builtin_amdgcn_s_sleep(32);
                if (__hip_atomic_load(&queue->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;   // (the producer gave up)
                if (wall_clock64() - t0 > 500000000ll) {   // 5 s at 100 MHz: give up, flag it
                    atomicExch(&queue->error, 1);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_unit = u;
    };
    for (;;) {
        if (threadIdx.x == 0) take_unit();
        __syncthreads();
        const long u = wave_uniform(s_unit);
        __syncthreads();
        if (u < 0) break;
        const long a0 = atom_offset[u];
        const int n = (int)(atom_offset[u + 1] - a0);
        const long v0 = a0 * vstride;
        PW_DCHECK(u < n_units && n >= 1 && n <= nmax, 100);
        PW_DCHECK(__builtin_amdgcn_read_exec() == ~0ull, 110);
        if (threadIdx.x == 0) ws->unit = u;
        int ncl = -1;
        analyse_unit<T, MASK_SAMPLING>(sh, ws, n, xyz + 3 * a0, vdw + v0, mass + v0, MASK_SAMPLING, out + u, prm,
                                       (FitTicket*)fa.tickets + u, &ncl);
        ncl = __builtin_amdgcn_readfirstlane(ncl);
        if ((fa.debug & 1) && ncl >= 1) ncl = PW_W_MAX + 1;
        // ticket and record were written by several waves: every wave releases its own stores, then thread 0
        // publishes one item per cluster (or lists the unit for the follow-up launch of the fused search),
        // counts the unit -- the team that counts the last one closes the queue -- and takes the next unit
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (ncl > PW_W_MAX) {
                const int at = atomicAdd(&fa.q->n_deferred, 1);
                fa.deferred[at] = (int)u;
            } else if (ncl >= 1) {
                const long pos = (long)atomicAdd(&fa.q->tail, (unsigned long long)ncl);
                for (int i = 0; i < ncl; ++i)
                    __hip_atomic_store(&fa.slots2[pos + i], (int)u * 16 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int done = atomicAdd(&fa.q->units_done, 1) + 1;
            if (done == (int)n_units) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const unsigned long long t = __hip_atomic_load(&fa.q->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&fa.q->final, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();       // (keeps this thread-0 region and the one at the top of the loop apart, see above)
    }
}

}  // namespace

// static LDS of the kernel (what a launch plan has to leave free beside its dynamic request)
extern "C" size_t pw_internal_sampling_static_lds(void) { return ((sizeof(UnitShared) + sizeof(pw_params) + 8 + 255) / 256) * 256; }

extern "C" int pw_internal_sampling_launch(void* stream, int grid, size_t lds_bytes, long n_units, const long* atom_offset,
                                           const double* xyz, const double* vdw, const double* mass, int nmax, int nrot, int nlb,
                                           const PwWsArgs* wsa, pw_unit_out* out, UnitQueue* queue, int* slots,
                                           const pw_params* prm, const unsigned* rsq_tab, int vstride, const FitArgs* fa) {
    // (the kernel's own team state -- UnitShared, parameters -- is static LDS: pw_internal_sampling_static_lds)
    // the limit, not a request: set once per device to the most a launch may ask for
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (e == hipSuccess && !(attr_done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)pw_sampling_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 256 - (int)pw_internal_sampling_static_lds());
        if (e == hipSuccess) attr_done.fetch_or(bit, std::memory_order_release);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pw_sampling_kernel, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, n_units, atom_offset, xyz,
                           vdw, mass, nmax, nrot, nlb, *wsa, out, queue, slots, *prm, rsq_tab, vstride, *fa);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        snprintf(pw_internal_error_buffer(), 512, "pw_sampling_kernel: %s", hipGetErrorString(e));
        return PW_E_HIP;
    }
    return PW_OK;
}
