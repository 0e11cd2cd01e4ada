// pw_shape.hip -- gfx950 kernels and C ABI entries of the shape descriptors and the circumcircle
// window estimate (include/pywindow_amd.h: pw_shape_batch, pw_circumcircle; reference
// utilities.py:434-650, 1653-1691).  One workgroup of four wavefronts per molecule; the N x N
// inertia sums are generated on the fly from the L2-resident coordinates, leaf by leaf of
// numpy's pairwise recursion (pw_shape.hpp).
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/pywindow_amd.h"
#include "pw_host.hpp"
#include "pw_shape.hpp"
#include "pw_team.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip
extern "C" int pw_context_device(pw_context* ctx);

namespace {

constexpr int SH_WAVES = 4;

__global__ void __launch_bounds__(SH_WAVES * 64)
pw_shape_kernel(long n_units, const long* __restrict__ off, const double* __restrict__ xyz,
                const double* __restrict__ mass, pw_shape_out* __restrict__ out, int mstride) {
    using T = DeviceTeam<SH_WAVES>;
    __shared__ ShapeScratch sc;
    for (long u = blockIdx.x; u < n_units; u += gridDim.x) {
        long a0 = off[u];
        shape_unit<T>(sc, xyz + 3 * a0, mass + a0 * mstride, (int)(off[u + 1] - a0), out + u);
    }
}

__global__ void __launch_bounds__(64)
pw_circumcircle_kernel(long n_sets, const double* __restrict__ xyz, const int* __restrict__ sets,
                       double* __restrict__ diameter, double* __restrict__ centre) {
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_sets) circumcircle_one(xyz, sets + 3 * k, diameter + k, centre + 3 * k);
}

struct Buffers {
    static constexpr int CAP = 8;
    void* p[CAP];
    int n = 0;
    ~Buffers() { for (int i = 0; i < n; ++i) if (p[i]) (void)hipFree(p[i]); }
    template <class X> hipError_t alloc(X** out, size_t bytes) {
        if (n >= CAP) return hipErrorOutOfMemory;
        hipError_t e = hipMalloc((void**)out, bytes ? bytes : 8);
        if (e == hipSuccess) p[n++] = *out;
        return e;
    }
};

}  // namespace

#define SH_TRY(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(pw_internal_error_buffer(), 512, "%s: %s", #call, hipGetErrorString(e_)); \
            return PW_E_HIP;                                                               \
        }                                                                                  \
    } while (0)

extern "C" int pw_shape_batch(pw_context* ctx, const pw_batch_in* in, pw_shape_out* out) {
    if (!ctx || !in || !out || in->n_units < 0 || (in->n_units && (!in->atom_offset || !in->xyz || !in->mass)))
        return PW_E_BAD_ARG;
    const long U = (long)in->n_units;
    if (U == 0) return PW_OK;
    PW_LOCK_CONTEXT(ctx);
    const long A = (long)in->atom_offset[U];
    const long TA = (long)in->template_atoms;       // > 0: one mass template for every unit
    if (TA < 0) return PW_E_BAD_ARG;
    for (long u = 0; u < U; ++u) {
        long nu = (long)(in->atom_offset[u + 1] - in->atom_offset[u]);
        if (nu <= 0 || nu > 46340 || (TA > 0 && nu != TA)) {
            snprintf(pw_internal_error_buffer(), 512, "pw_shape_batch: unit %ld is empty, too large or not the template's size", u);
            return PW_E_BAD_ARG;
        }
    }
    DeviceScope dev_scope_;
    if (pw_context_device(ctx) < 0) {
        snprintf(pw_internal_error_buffer(), 512, "not part of the host path (device = -1 runs the analysis only)");
        return PW_E_NO_DEVICE;
    }
    SH_TRY(dev_scope_.enter(pw_context_device(ctx)));
    hipStream_t st = (hipStream_t)pw_context_stream(ctx);
    Buffers buf;
    long* d_off;
    double *d_xyz, *d_mass;
    pw_shape_out* d_out;
    SH_TRY(buf.alloc(&d_off, sizeof(long) * (U + 1)));
    SH_TRY(buf.alloc(&d_xyz, sizeof(double) * 3 * A));
    SH_TRY(buf.alloc(&d_mass, sizeof(double) * (TA > 0 ? TA : A)));
    SH_TRY(buf.alloc(&d_out, sizeof(pw_shape_out) * U));
    SH_TRY(hipMemcpyAsync(d_off, in->atom_offset, sizeof(long) * (U + 1), hipMemcpyHostToDevice, st));
    SH_TRY(hipMemcpyAsync(d_xyz, in->xyz, sizeof(double) * 3 * A, hipMemcpyHostToDevice, st));
    SH_TRY(hipMemcpyAsync(d_mass, in->mass, sizeof(double) * (TA > 0 ? TA : A), hipMemcpyHostToDevice, st));
    long grid = U < 4096 ? U : 4096;
    hipLaunchKernelGGL(pw_shape_kernel, dim3((unsigned)grid), dim3(SH_WAVES * 64), 0, st, U, d_off, d_xyz, d_mass, d_out,
                       TA > 0 ? 0 : 1);
    SH_TRY(hipGetLastError());
    SH_TRY(hipMemcpyAsync(out, d_out, sizeof(pw_shape_out) * U, hipMemcpyDeviceToHost, st));
    SH_TRY(hipStreamSynchronize(st));
    return PW_OK;
}

extern "C" int pw_circumcircle(pw_context* ctx, const double* xyz, int64_t n_atoms, const int32_t* atom_sets,
                               int64_t n_sets, double* diameter, double* centre) {
    if (!ctx || n_sets < 0 || n_atoms <= 0 || !xyz || (n_sets && (!atom_sets || !diameter || !centre)))
        return PW_E_BAD_ARG;
    if (n_sets == 0) return PW_OK;
    PW_LOCK_CONTEXT(ctx);
    for (long k = 0; k < 3 * (long)n_sets; ++k)
        if (atom_sets[k] < 0 || atom_sets[k] >= n_atoms) {
            snprintf(pw_internal_error_buffer(), 512, "pw_circumcircle: atom index %d out of range", atom_sets[k]);
            return PW_E_BAD_ARG;       // the reference: IndexError
        }
    DeviceScope dev_scope_;
    if (pw_context_device(ctx) < 0) {
        snprintf(pw_internal_error_buffer(), 512, "not part of the host path (device = -1 runs the analysis only)");
        return PW_E_NO_DEVICE;
    }
    SH_TRY(dev_scope_.enter(pw_context_device(ctx)));
    hipStream_t st = (hipStream_t)pw_context_stream(ctx);
    Buffers buf;
    double *d_xyz, *d_d, *d_c;
    int* d_sets;
    SH_TRY(buf.alloc(&d_xyz, sizeof(double) * 3 * n_atoms));
    SH_TRY(buf.alloc(&d_sets, sizeof(int) * 3 * n_sets));
    SH_TRY(buf.alloc(&d_d, sizeof(double) * n_sets));
    SH_TRY(buf.alloc(&d_c, sizeof(double) * 3 * n_sets));
    SH_TRY(hipMemcpyAsync(d_xyz, xyz, sizeof(double) * 3 * n_atoms, hipMemcpyHostToDevice, st));
    SH_TRY(hipMemcpyAsync(d_sets, atom_sets, sizeof(int) * 3 * n_sets, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pw_circumcircle_kernel, dim3((unsigned)((n_sets + 63) / 64)), dim3(64), 0, st, (long)n_sets,
                       d_xyz, d_sets, d_d, d_c);
    SH_TRY(hipGetLastError());
    SH_TRY(hipMemcpyAsync(diameter, d_d, sizeof(double) * n_sets, hipMemcpyDeviceToHost, st));
    SH_TRY(hipMemcpyAsync(centre, d_c, sizeof(double) * 3 * n_sets, hipMemcpyDeviceToHost, st));
    SH_TRY(hipStreamSynchronize(st));
    return PW_OK;
}
