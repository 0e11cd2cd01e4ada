// pw_team.hpp -- the execution-team abstraction the unit pipeline is written
// against.  One *team* analyses one (frame, molecule) unit.
//
//   * DeviceTeam<NW>: a workgroup of NW wavefronts (64 lanes each) on gfx950.
//     Wave-level reductions are exact: min/max/argmin through DPP moves; the
//     only floating-point sums that cross lanes are folded in numpy's fixed
//     order (np_leaf_sums_t), so results do not depend on the team shape.
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
    PW_HD static void row_argmin4(double v, int idx, double* outv, int* outi) {
        for (int r = 0; r < 4; ++r) { outv[r] = v; outi[r] = idx; }
    }
    PW_HD static double wave_min(double v) { return v; }
    PW_HD static void row_min4(double v, double* outv) { for (int r = 0; r < 4; ++r) outv[r] = v; }
    PW_HD static unsigned long long ballot(bool p) { return p ? 1ull : 0ull; }
    PW_HD static bool wave_all(bool p) { return p; }
    PW_HD static bool wave_any(bool p) { return p; }
    PW_HD static double xor_d(double v, int /*lane_mask*/) { return v; }
    PW_HD static double bcast(double v, int /*src_lane*/) { return v; }
    PW_HD static double bcast_u(double v, int /*uniform_src_lane*/) { return v; }
    PW_HD static int bcast_i(int v, int /*src_lane*/) { return v; }
    PW_HD static int shfl_up_i(int v, int /*delta*/) { return v; }
    PW_HD static int uniform_i(int v) { return v; }
};

#if defined(__HIPCC__)
#if defined(PW_BARRIER_PROF)
// Diagnostic build (-DPW_PROFILE -DPW_BARRIER_PROF, tests/tools/profile_barriers.py): what every wave of a team waits at
// the team's barriers, 100 MHz ticks, collected per wave and handed to the stage timer that closes next (PW_T1).
__shared__ unsigned long long pw_bar_acc[8];
#endif
template <int NW>
struct DeviceTeam {
    static constexpr int NWAVES = NW;
    static constexpr int WSIZE = 64;
    static constexpr int SIZE = NW * 64;
    __device__ static int tid() { return threadIdx.x; }
    __device__ static int lane() { return threadIdx.x & 63; }
    __device__ static int wave() { return threadIdx.x >> 6; }
#if defined(PW_BARRIER_PROF)
    __device__ static void sync() {
        const long long t0 = wall_clock64();
        __syncthreads();
        if (lane() == 0) atomicAdd(&pw_bar_acc[wave()], (unsigned long long)(wall_clock64() - t0));
    }
#else
    __device__ static void sync() { __syncthreads(); }
#endif
    // waves execute in lockstep; LDS traffic inside one wave only needs the
    // compiler not to reorder across this point and the LDS queue drained
    __device__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#if defined(PW_GENERIC_TEAM_MEM)
        // team memory is global here: lanes exchange data through the vector memory path, so its queue
        // is drained as well
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- DPP reductions -----------------------------------------------------------
    // Cross-lane moves through the data-parallel-primitive path (no LDS crossbar):
    // row_shr 1,2,4,8 fold each row of 16 lanes into its lane 15, row_bcast15/31 fold
    // the four rows into lane 63, v_readlane broadcasts.  Lanes that receive nothing
    // keep their own value, so op(x, x) = x for min/max.
    template <int CTRL, int ROW_MASK, int BANK_MASK>
    __device__ static int dpp_i(int x) {
        return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, BANK_MASK, false);
    }
    template <int CTRL, int ROW_MASK, int BANK_MASK>
    __device__ static double dpp_d(double x) {
        union { double d; int i[2]; } a, b;
        a.d = x;
        b.i[0] = dpp_i<CTRL, ROW_MASK, BANK_MASK>(a.i[0]);
        b.i[1] = dpp_i<CTRL, ROW_MASK, BANK_MASK>(a.i[1]);
        return b.d;
    }
    __device__ static double lane_d(double x, int lane) {
        union { double d; int i[2]; } a;
        a.d = x;
        a.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
        a.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
        return a.d;
    }
    template <bool IS_MIN, int CTRL, int ROW_MASK, int BANK_MASK>
    __device__ static void arg_step(double& v, int& idx) {
        double ov = dpp_d<CTRL, ROW_MASK, BANK_MASK>(v);
        int oi = dpp_i<CTRL, ROW_MASK, BANK_MASK>(idx);
        bool take = IS_MIN ? ((ov < v) || (ov == v && oi < idx)) : ((ov > v) || (ov == v && oi < idx));
        v = take ? ov : v;
        idx = take ? oi : idx;
    }
    template <bool IS_MIN>
    __device__ static void arg_reduce16(double& v, int& idx) {   // result in lane 15 of each row
        arg_step<IS_MIN, 0x111, 0xf, 0xf>(v, idx);
        arg_step<IS_MIN, 0x112, 0xf, 0xf>(v, idx);
        arg_step<IS_MIN, 0x114, 0xf, 0xe>(v, idx);
        arg_step<IS_MIN, 0x118, 0xf, 0xc>(v, idx);
    }
    template <bool IS_MIN>
    __device__ static void arg_reduce64(double& v, int& idx) {
        arg_reduce16<IS_MIN>(v, idx);
        arg_step<IS_MIN, 0x142, 0xa, 0xf>(v, idx);
        arg_step<IS_MIN, 0x143, 0xc, 0xf>(v, idx);
        v = lane_d(v, 63);
        idx = __builtin_amdgcn_readlane(idx, 63);
    }
    __device__ static void wave_argmin(double& v, int& idx) { arg_reduce64<true>(v, idx); }
    __device__ static void wave_argmax(double& v, int& idx) { arg_reduce64<false>(v, idx); }
    // four independent reductions, one per row of 16 lanes; out[r] = result of row r (all lanes)
    __device__ static void row_argmin4(double v, int idx, double* outv, int* outi) {
        arg_reduce16<true>(v, idx);
        outv[0] = lane_d(v, 15); outi[0] = __builtin_amdgcn_readlane(idx, 15);
        outv[1] = lane_d(v, 31); outi[1] = __builtin_amdgcn_readlane(idx, 31);
        outv[2] = lane_d(v, 47); outi[2] = __builtin_amdgcn_readlane(idx, 47);
        outv[3] = lane_d(v, 63); outi[3] = __builtin_amdgcn_readlane(idx, 63);
    }
    // value-only minimum: the same DPP ladder without the index
    template <int CTRL, int ROW_MASK, int BANK_MASK>
    __device__ static void min_step(double& v) {
        v = __builtin_fmin(v, dpp_d<CTRL, ROW_MASK, BANK_MASK>(v));
    }
    __device__ static double wave_min(double v) {
        min_step<0x111, 0xf, 0xf>(v);
        min_step<0x112, 0xf, 0xf>(v);
        min_step<0x114, 0xf, 0xe>(v);
        min_step<0x118, 0xf, 0xc>(v);
        min_step<0x142, 0xa, 0xf>(v);
        min_step<0x143, 0xc, 0xf>(v);
        return lane_d(v, 63);
    }
    // four independent value-only minima, one per row of 16 lanes
    __device__ static void row_min4(double v, double* outv) {
        min_step<0x111, 0xf, 0xf>(v);
        min_step<0x112, 0xf, 0xf>(v);
        min_step<0x114, 0xf, 0xe>(v);
        min_step<0x118, 0xf, 0xc>(v);
        outv[0] = lane_d(v, 15); outv[1] = lane_d(v, 31); outv[2] = lane_d(v, 47); outv[3] = lane_d(v, 63);
    }
    __device__ static unsigned long long ballot(bool p) { return __ballot(p); }
    __device__ static bool wave_all(bool p) { return __all(p); }
    __device__ static bool wave_any(bool p) { return __any(p); }
    __device__ static double xor_d(double v, int lane_mask) { return __shfl_xor(v, lane_mask, 64); }
    __device__ static double bcast(double v, int src) { return __shfl(v, src, 64); }
    // source lane known to be the same in every lane: v_readlane, no LDS crossbar
    __device__ static double bcast_u(double v, int src) { return lane_d(v, __builtin_amdgcn_readfirstlane(src)); }
    __device__ static int bcast_i(int v, int src) { return __shfl(v, src, 64); }
    __device__ static int shfl_up_i(int v, int delta) { return __shfl_up(v, delta, 64); }
    // a value every lane of the wave holds alike, as the scalar it is (loop bounds and branches on it are scalar)
    __device__ static int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
    // inclusive prefix sum over the lanes of the wave: four row_shr steps inside every row of 16 lanes (a lane
    // without a source adds 0), then the totals of the rows below by v_readlane
    __device__ static int incl_scan_i(int v) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
        const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31),
                  r2 = __builtin_amdgcn_readlane(v, 47);
        const int row = (threadIdx.x & 63) >> 4;
        return v + (row >= 1 ? r0 : 0) + (row >= 2 ? r1 : 0) + (row >= 3 ? r2 : 0);
    }
};
#endif

}  // namespace pw
