// pw_team.hpp -- the execution-team abstraction the unit pipeline is written
// against.  One *team* analyses one (frame, molecule) unit.
//
//   * DeviceTeam<NW>: a workgroup of NW wavefronts (64 lanes each) on gfx950.
//     Wave-level reductions are DPP/ds_swizzle-free __shfl exchanges (exact:
//     min/max/argmin only -- no floating-point sums are ever reduced across
//     lanes, so results do not depend on the team shape).
//   * HostTeam: one thread; used by tests/hostsim to run the identical
//     pipeline on a CPU without a GPU.
#pragma once
#include "pw_common.hpp"

namespace pw {

struct HostTeam {
    static constexpr int NWAVES = 1;
    static constexpr int WSIZE = 1;
    static constexpr int SIZE = 1;
    PW_HD static int tid() { return 0; }
    PW_HD static int lane() { return 0; }
    PW_HD static int wave() { return 0; }
    PW_HD static void sync() {}
    PW_HD static void wave_sync() {}
    // (value, index) minimum with smallest-index tie-break, broadcast to all lanes
    PW_HD static void wave_argmin(double& v, int& idx) { (void)v; (void)idx; }
    PW_HD static void wave_argmax(double& v, int& idx) { (void)v; (void)idx; }
    PW_HD static unsigned long long ballot(bool p) { return p ? 1ull : 0ull; }
    PW_HD static bool wave_all(bool p) { return p; }
    PW_HD static bool wave_any(bool p) { return p; }
    PW_HD static double bcast(double v, int /*src_lane*/) { return v; }
    PW_HD static int bcast_i(int v, int /*src_lane*/) { return v; }
};

#if defined(__HIPCC__)
template <int NW>
struct DeviceTeam {
    static constexpr int NWAVES = NW;
    static constexpr int WSIZE = 64;
    static constexpr int SIZE = NW * 64;
    __device__ static int tid() { return threadIdx.x; }
    __device__ static int lane() { return threadIdx.x & 63; }
    __device__ static int wave() { return threadIdx.x >> 6; }
    __device__ static void sync() { __syncthreads(); }
    // waves execute in lockstep; LDS traffic inside one wave only needs the
    // compiler not to reorder across this point and the LDS queue drained
    __device__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ static void wave_argmin(double& v, int& idx) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(v, off, 64);
            int oi = __shfl_xor(idx, off, 64);
            bool take = (ov < v) || (ov == v && oi < idx);
            v = take ? ov : v;
            idx = take ? oi : idx;
        }
    }
    __device__ static void wave_argmax(double& v, int& idx) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            double ov = __shfl_xor(v, off, 64);
            int oi = __shfl_xor(idx, off, 64);
            bool take = (ov > v) || (ov == v && oi < idx);
            v = take ? ov : v;
            idx = take ? oi : idx;
        }
    }
    __device__ static unsigned long long ballot(bool p) { return __ballot(p); }
    __device__ static bool wave_all(bool p) { return __all(p); }
    __device__ static bool wave_any(bool p) { return __any(p); }
    __device__ static double bcast(double v, int src) { return __shfl(v, src, 64); }
    __device__ static int bcast_i(int v, int src) { return __shfl(v, src, 64); }
};
#endif

}  // namespace pw
