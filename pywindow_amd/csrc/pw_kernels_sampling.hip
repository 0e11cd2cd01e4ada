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
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/pywindow_amd.h"
#include "pw_unit.hpp"
#include "pw_launch.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip (512 bytes, thread local)

#ifndef PW_OCC_S
#define PW_OCC_S 3
#endif

namespace {

constexpr unsigned MASK_SAMPLING = PW_STAGE_WIN_BULK | PW_STAGE_REUSE_OPT | PW_STAGE_MERGE | PW_STAGE_COM_ONLY;

__global__ void __launch_bounds__(256, PW_OCC_S)
pw_sampling_kernel(long n_units, const long* __restrict__ atom_offset, const double* __restrict__ xyz,
                   const double* __restrict__ vdw, const double* __restrict__ mass, int nmax, int nrot, int nlb,
                   PwWsArgs wsa, pw_unit_out* __restrict__ out, UnitQueue* queue, int* __restrict__ slots, pw_params prm,
                   const unsigned* __restrict__ rsq_tab, int vstride, FitArgs fa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ long s_unit;
    using T = DeviceTeam<4>;
    UnitShared sh;
    // one frame, shifted in place; the DBSCAN bit sets but no per-cluster arrays (they are the ticket's); the
    // window frames and optimiser blocks a fused team would carve are the scratch arena of the sampling stages
    sh.carve(lds, nmax, nrot, nlb, 1, 2, wsa.p_cap);
    TeamWorkspace* ws = (TeamWorkspace*)wsa.ws + blockIdx.x;
    if (threadIdx.x == 0) bind_workspace(ws, wsa, blockIdx.x, rsq_tab, team_slab_bytes(wsa.p_cap), team_adj_words(wsa.p_cap));
    __syncthreads();
    for (;;) {
        // consumer of the optimiser launch: ONE relaxed poll loop -> agent acquire -> team barrier -> loads
        if (threadIdx.x == 0) {
            long pos = (long)atomicAdd(&queue->head, 1ull);
            long u = -1;
            if (pos < n_units) {
                long long t0 = wall_clock64();
                for (;;) {
                    int v = __hip_atomic_load(&slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v >= 0) { u = v; break; }
                    __builtin_amdgcn_s_sleep(32);
                    if (wall_clock64() - t0 > 500000000ll) {   // 5 s at 100 MHz: give up, flag it
                        atomicExch(&queue->error, 1);
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s_unit = u;
        }
        __syncthreads();
        const long u = s_unit;
        __syncthreads();
        if (u < 0) break;
        const long a0 = atom_offset[u];
        const int n = (int)(atom_offset[u + 1] - a0);
        const long v0 = a0 * vstride;
        if (threadIdx.x == 0) ws->unit = u;
        int ncl = -1;
        analyse_unit<T, MASK_SAMPLING>(sh, ws, n, xyz + 3 * a0, vdw + v0, mass + v0, MASK_SAMPLING, out + u, prm,
                                       (FitTicket*)fa.tickets + u, &ncl);
        // ticket and record were written by several waves: every wave releases its own stores, then thread 0
        // publishes one item per cluster (or lists the unit for the follow-up launch of the fused search) and
        // counts the unit; the team that counts the last one closes the queue
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
    }
}

}  // namespace

extern "C" int pw_internal_sampling_launch(void* stream, int grid, size_t lds_bytes, long n_units, const long* atom_offset,
                                           const double* xyz, const double* vdw, const double* mass, int nmax, int nrot, int nlb,
                                           const PwWsArgs* wsa, pw_unit_out* out, UnitQueue* queue, int* slots,
                                           const pw_params* prm, const unsigned* rsq_tab, int vstride, const FitArgs* fa) {
    // the limit, not a request (set on every call: cheap, and valid for whichever device is current)
    hipError_t e = hipFuncSetAttribute((const void*)pw_sampling_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
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
