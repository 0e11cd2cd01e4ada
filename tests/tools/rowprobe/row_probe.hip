// Prototype (measurement only): four optimiser chains per wavefront, one per row of 16 lanes.
// Same Lbfgsb<3> source as the product, instantiated for a 16-lane team.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../../pywindow_amd/csrc/pw_unit.hpp"
using namespace pw;

struct RowTeam {
    static constexpr int NWAVES = 1;
    static constexpr int WSIZE = 16;
    static constexpr int SIZE = 16;
    __device__ static int tid() { return threadIdx.x & 15; }
    __device__ static int lane() { return threadIdx.x & 15; }
    __device__ static int wave() { return 0; }
    __device__ static void sync() { wave_sync(); }
    __device__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ static double bcast_u(double v, int src) { return __shfl(v, src, 16); }
};

#ifndef ROW_OCC
#define ROW_OCC 2
#endif

// value-only minimum over a row of 16 lanes (DPP row shifts), result broadcast to the row
__device__ inline double row_min(double v) {
    using D = DeviceTeam<1>;
    D::min_step<0x111, 0xf, 0xf>(v);
    D::min_step<0x112, 0xf, 0xf>(v);
    D::min_step<0x114, 0xf, 0xe>(v);
    D::min_step<0x118, 0xf, 0xc>(v);
    return __shfl(v, 15, 16);
}

struct RowIn { double x0[3]; double r; };
struct RowOut { double x[3]; double f; int nit, nfev, task, pad; };

template <int DEFER>
__global__ void __launch_bounds__(64, ROW_OCC) row_chain_kernel(long n_units, int n, const double* __restrict__ xyz,
                                                                const double* __restrict__ vdw, const RowIn* __restrict__ in,
                                                                RowOut* __restrict__ out, unsigned long long* counter) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using T = RowTeam;
    const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
    const size_t nn = (size_t)((n + 1) & ~1);
    const size_t row_bytes = nn * 8 * 5 + ((sizeof(LbMem<3>) + 15) & ~(size_t)15);
    PW_LDS unsigned char* base = (PW_LDS unsigned char*)lds + row * row_bytes;
    Frame F;
    F.x = (ldouble*)base; F.y = F.x + nn; F.z = F.y + nn; F.xx = F.z + nn;
    ldouble* rv = F.xx + nn;
    F.vdw = rv; F.perm = nullptr; F.cls = nullptr;
    LbMem<3>* Smem = (LbMem<3>*)(rv + nn);
    Lbfgsb<3> S;
    int state = 0;      // 0 idle, 1 running, 2 done
    long unit = -1;
    double lo[3], up[3];
    int nit = 0, nfev = 0, waited = 0;
    bool have_last = false;
    double lx = 0, ly = 0, lz = 0, lf = 0, lg0 = 0, lg1 = 0, lg2 = 0;
    S.task = LB_STOP;
    for (;;) {
        if (state == 0) {
            long u = -1;
            if (l == 0) u = (long)atomicAdd(counter, 1ull);
            u = __shfl(u, 0, 16);
            if (u >= n_units) state = 2;
            else {
                unit = u;
                const double* c = xyz + 3 * (size_t)n * u;
                for (int i = l; i < n; i += 16) {
                    double x = c[3 * i], y = c[3 * i + 1], z = c[3 * i + 2];
                    F.x[i] = x; F.y[i] = y; F.z[i] = z; F.xx[i] = sq3(x, y, z); rv[i] = vdw[i];
                }
                double x0[3];
                int nbd[3] = {2, 2, 2};
                for (int k = 0; k < 3; ++k) { x0[k] = in[u].x0[k]; lo[k] = x0[k] - in[u].r; up[k] = x0[k] + in[u].r; }
                T::wave_sync();
                S.template setup<T>(Smem, x0, lo, up, nbd, 1e7, 1e-5, 20);
                nit = 0; nfev = 0; have_last = false; waited = 0;
                state = 1;
            }
        }
        if (__all(state == 2)) break;
        if (state == 1) {
            bool go = true;
            if (S.task == LB_FG) {
                double px = S.x[0], py = S.x[1], pz = S.x[2];
                if (!(have_last && px == lx && py == ly && pz == lz)) {
                    double qx[4] = {px, px, px, px}, qy[4] = {py, py, py, py}, qz[4] = {pz, pz, pz, pz}, dxs[3];
                    for (int c = 0; c < 3; ++c) {
                        double xc = c == 0 ? px : (c == 1 ? py : pz);
                        double h = fd_step(xc, lo[c], up[c]);
                        double x1 = xc + h;
                        dxs[c] = x1 - xc;
                        if (c == 0) qx[1] = x1; else if (c == 1) qy[2] = x1; else qz[3] = x1;
                    }
                    double pp[4], best[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { pp[q] = sq3(qx[q], qy[q], qz[q]); best[q] = PW_INF; }
                    for (int i = l; i < n; i += 16) {
                        const double ax = F.x[i], ay = F.y[i], az = F.z[i], aq = F.xx[i], ar = rv[i];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            double g = pw_fma(az, qz[q], pw_fma(ax, qx[q], ay * qy[q]));
                            double d2 = pw_m2add(g, aq) + pp[q];
                            double d = pw_sqrt(d2 > 0.0 ? d2 : 0.0);
                            best[q] = __builtin_fmin(best[q], d - ar);
                        }
                    }
                    double gv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gv[q] = row_min(best[q]);
                    double f0 = -(gv[0] * 2.0);
                    lg0 = (-(gv[1] * 2.0) - f0) / dxs[0];
                    lg1 = (-(gv[2] * 2.0) - f0) / dxs[1];
                    lg2 = (-(gv[3] * 2.0) - f0) / dxs[2];
                    lf = f0; lx = px; ly = py; lz = pz;
                    have_last = true;
                    nfev += 4;
                }
                S.f = lf;
                S.g[0] = lg0; S.g[1] = lg1; S.g[2] = lg2;
                T::wave_sync();
            } else if (S.task == LB_NEW_X) {
                if (DEFER) {
                    // an iteration (the long path of step()) is run when at least DEFER rows want one, or no
                    // row of the wave is in a line search, or this row has waited long enough
                    unsigned long long want = __ballot(true);          // rows here: task == NEW_X
                    unsigned long long running = __ballot(true);
                    (void)running;
                    int rows_want = __popcll(want) >> 4;
                    go = rows_want >= DEFER || waited >= 6;
                    waited = go ? 0 : waited + 1;
                }
                if (go) {
                    nit += 1;
                    if (nit >= 15000 || nfev > 15000) S.task = LB_STOP;
                }
            }
            if (go) {
                S.template step<T>();
                T::wave_sync();
            }
            if (S.task != LB_FG && S.task != LB_NEW_X) {
                if (l == 0) {
                    RowOut& o = out[unit];
                    o.x[0] = S.x[0]; o.x[1] = S.x[1]; o.x[2] = S.x[2]; o.f = S.f; o.nit = nit; o.nfev = nfev; o.task = S.task;
                }
                state = 0;
            }
        }
    }
}

extern "C" int row_probe_run(long n_units, int n, const double* xyz, const double* vdw, const RowIn* in, RowOut* out,
                             int defer, int iters, float* ms) {
    double *d_xyz, *d_vdw;
    RowIn* d_in;
    RowOut* d_out;
    unsigned long long* d_c;
    hipMalloc(&d_xyz, sizeof(double) * 3 * n * n_units);
    hipMalloc(&d_vdw, sizeof(double) * n);
    hipMalloc(&d_in, sizeof(RowIn) * n_units);
    hipMalloc(&d_out, sizeof(RowOut) * n_units);
    hipMalloc(&d_c, 8);
    hipMemcpy(d_xyz, xyz, sizeof(double) * 3 * n * n_units, hipMemcpyHostToDevice);
    hipMemcpy(d_vdw, vdw, sizeof(double) * n, hipMemcpyHostToDevice);
    hipMemcpy(d_in, in, sizeof(RowIn) * n_units, hipMemcpyHostToDevice);
    const size_t nn = (size_t)((n + 1) & ~1);
    const size_t lds = 4 * (nn * 8 * 5 + ((sizeof(LbMem<3>) + 15) & ~(size_t)15));
    auto k0 = row_chain_kernel<0>;
    auto k2 = row_chain_kernel<2>;
    auto k3 = row_chain_kernel<3>;
    auto k4 = row_chain_kernel<4>;
    auto kern = defer == 0 ? k0 : (defer == 2 ? k2 : (defer == 3 ? k3 : k4));
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    int grid = (int)((n_units + 3) / 4);
    if (grid > 2048) grid = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < iters + 1; ++it) {
        hipMemset(d_c, 0, 8);
        if (it == 1) hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, 0, n_units, n, d_xyz, d_vdw, d_in, d_out, d_c);
    }
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { fprintf(stderr, "row probe: %s\n", hipGetErrorString(e)); return -1; }
    hipEventElapsedTime(ms, e0, e1);
    *ms /= iters;
    hipMemcpy(out, d_out, sizeof(RowOut) * n_units, hipMemcpyDeviceToHost);
    hipFree(d_xyz); hipFree(d_vdw); hipFree(d_in); hipFree(d_out); hipFree(d_c);
    return 0;
}
