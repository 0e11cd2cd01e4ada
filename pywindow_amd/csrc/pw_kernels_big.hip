// pw_kernels_big.hip -- the analysis for molecules that do not fit a CU's LDS.
//
// The reference has no upper limit on the number of atoms (utilities.py works on whatever arrays it is
// given).  The kernels of pw_kernels.hip keep a unit's coordinates in LDS for its whole lifetime, which
// holds about 1700 atoms; beyond that this translation unit takes over: THE SAME SOURCE (pw_unit.hpp), compiled
// with PW_GENERIC_TEAM_MEM so that the team's shared block -- coordinates, radii, optimiser states, bit
// sets -- is a slab of global memory (L2-resident) instead of LDS, behind unqualified pointers.  Slower per
// atom than the LDS path and only ever used for such molecules; results are the same bits (the arithmetic
// does not know where its operands live).
#define PW_GENERIC_TEAM_MEM 1
#define pw pw_big          // a namespace of its own: its inline functions must not merge with pw_kernels.hip's
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/pywindow_amd.h"
#include "pw_unit.hpp"
#include "pw_launch.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip (512 bytes, thread local)

namespace {

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2)
pw_analyse_big_kernel(long n_units, const long* __restrict__ atom_offset, const double* __restrict__ xyz,
                      const double* __restrict__ vdw, const double* __restrict__ mass, unsigned stages, int nmax,
                      PwWsArgs wsa, unsigned char* __restrict__ blocks, size_t block_bytes, unsigned long long* counter,
                      pw_unit_out* __restrict__ out, pw_params prm, const unsigned* __restrict__ rsq_tab, int vstride) {
    __shared__ long s_unit;
    using T = DeviceTeam<NW>;
    UnitShared sh;
    sh.carve(blocks + (size_t)blockIdx.x * block_bytes, nmax, 4, 4, 2, false, wsa.p_cap);
    TeamWorkspace* ws = (TeamWorkspace*)wsa.ws + blockIdx.x;
    if (threadIdx.x == 0) bind_workspace(ws, wsa, blockIdx.x, rsq_tab, team_slab_bytes(wsa.p_cap), team_adj_words(wsa.p_cap));
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) {
            long u = (long)atomicAdd(counter, 1ull);
            s_unit = u < n_units ? u : -1;
        }
        __syncthreads();
        long u = s_unit;
        __syncthreads();
        if (u < 0) break;
        long a0 = atom_offset[u];
        int n = (int)(atom_offset[u + 1] - a0);
        const long v0 = a0 * vstride;
        if (threadIdx.x == 0) ws->unit = u;
        analyse_unit<T, 0xffffffffu>(sh, ws, n, xyz + 3 * a0, vdw + v0, mass + v0, stages, out + u, prm);
    }
}

}  // namespace

// bytes of one team's shared block (four window-fit slots, two coordinate frames)
extern "C" size_t pw_internal_big_block_bytes(int nmax, int p_cap) {
    return (UnitShared::bytes(nmax, 4, 4, 2, false, p_cap) + 255) & ~(size_t)255;
}

extern "C" int pw_internal_big_launch(void* stream, int grid, long n_units, const long* atom_offset, const double* xyz,
                                      const double* vdw, const double* mass, unsigned stages, int nmax, const PwWsArgs* wsa,
                                      unsigned char* blocks, size_t block_bytes, unsigned long long* counter, pw_unit_out* out,
                                      const pw_params* prm, const unsigned* rsq_tab, int vstride) {
    hipLaunchKernelGGL(pw_analyse_big_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_units, atom_offset, xyz, vdw,
                       mass, stages, nmax, *wsa, blocks, block_bytes, counter, out, *prm, rsq_tab, vstride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(pw_internal_error_buffer(), 512, "pw_analyse_big_kernel: %s", hipGetErrorString(e));
        return PW_E_HIP;
    }
    return PW_OK;
}
