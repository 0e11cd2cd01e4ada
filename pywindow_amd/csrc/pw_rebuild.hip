// pw_rebuild.hip -- gfx950 kernel and C ABI entry of the periodic pre-processing
// (include/pywindow_amd.h: pw_discrete_molecules; reference utilities.py:768-1085).
//
// One persistent workgroup (4 wavefronts) re-assembles one frame at a time; frames are
// handed out by an atomic counter.  All state of a frame lives in a per-team slab of
// global memory that stays L2-resident (value coordinates of the 27 images, candidate
// lists, visit stamps); the traversal itself is a chain of short dependent steps, so the
// launch is sized for many concurrent frames rather than for wide teams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pywindow_amd.h"
#include "pw_rebuild.hpp"
#include "pw_team.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip (512 bytes, thread local)
extern "C" int pw_context_device(pw_context* ctx);

namespace {

constexpr int RB_WAVES = 4;

__global__ void __launch_bounds__(RB_WAVES * 64)
pw_rebuild_kernel(pw_cell_in in, pw_cell_out out, unsigned char* __restrict__ slabs, size_t slab_bytes,
                  unsigned long long* counter) {
    using T = DeviceTeam<RB_WAVES>;
    __shared__ long s_frame;
    const int n = in.n_atoms;
    RebuildWs* w = RebuildWs::carve(slabs + (size_t)blockIdx.x * slab_bytes, n, in.rebuild, T::SIZE);
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) {
            long f = (long)atomicAdd(counter, 1ull);
            s_frame = f < in.n_frames ? f : -1;
        }
        __syncthreads();
        long f = s_frame;
        __syncthreads();
        if (f < 0) break;
        RebuildFrame fr;
        fr.n = n;
        fr.periodic = in.lattice != nullptr;
        fr.rebuild = in.rebuild;
        fr.xyz = in.xyz + (size_t)f * n * 3;
        fr.lattice = in.lattice ? in.lattice + 9 * f : nullptr;
        fr.lattice_inv = in.lattice_inv ? in.lattice_inv + 9 * f : nullptr;
        fr.cov = in.cov;
        fr.mass = in.mass;
        fr.terminal = in.terminal;
        fr.max_dist = in.max_dist;
        fr.tol = in.tol;
        RebuildOut o;
        o.n_mol = out.n_mol + f;
        o.status = out.status + f;
        o.mol_offset = out.mol_offset + (size_t)f * (out.mols_cap + 1);
        o.src_atom = out.src_atom + (size_t)f * out.atoms_cap;
        o.src_image = out.src_image + (size_t)f * out.atoms_cap;
        o.xyz = out.xyz + (size_t)f * out.atoms_cap * 3;
        o.atoms_cap = out.atoms_cap;
        o.mols_cap = out.mols_cap;
        // carve() ran before the loop with this team's pointers; work/work_next may have been
        // swapped by the previous frame, which is harmless (both are id-sized)
        rebuild_frame<T>(fr, *w, o);
    }
}

struct Buffers {
    void* p[16];
    int n = 0;
    ~Buffers() { for (int i = 0; i < n; ++i) if (p[i]) (void)hipFree(p[i]); }
    template <class X> hipError_t alloc(X** out, size_t bytes) {
        hipError_t e = hipMalloc((void**)out, bytes ? bytes : 8);
        if (e == hipSuccess) p[n++] = *out;
        return e;
    }
};

}  // namespace

#define RB_TRY(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(pw_internal_error_buffer(), 512, "%s: %s", #call, hipGetErrorString(e_)); \
            return PW_E_HIP;                                                               \
        }                                                                                  \
    } while (0)

extern "C" int pw_discrete_molecules(pw_context* ctx, const pw_cell_in* in, const pw_cell_out* out) {
    if (!ctx || !in || !out || in->n_frames < 0 || in->n_atoms <= 0 || !in->xyz || !in->cov || !in->mass ||
        !in->terminal || !out->n_mol || !out->status || !out->mol_offset || !out->src_atom ||
        !out->src_image || !out->xyz || out->atoms_cap <= 0 || out->mols_cap <= 0)
        return PW_E_BAD_ARG;
    if (in->rebuild && (!in->lattice || !in->lattice_inv)) {
        snprintf(pw_internal_error_buffer(), 512, "rebuild needs the lattice and its inverse");
        return PW_E_BAD_ARG;
    }
    if (in->lattice && !in->lattice_inv) return PW_E_BAD_ARG;
    if (in->n_frames == 0) return PW_OK;
    RB_TRY(hipSetDevice(pw_context_device(ctx)));
    hipStream_t st = (hipStream_t)pw_context_stream(ctx);
    const long F = (long)in->n_frames;
    const int n = in->n_atoms;
    hipDeviceProp_t prop;
    RB_TRY(hipGetDeviceProperties(&prop, pw_context_device(ctx)));
    size_t slab = (RebuildWs::bytes(n, in->rebuild, RB_WAVES * 64) + 255) & ~(size_t)255;
    long grid = (long)prop.multiProcessorCount * 4;
    if (grid > F) grid = F;
    // keep the slabs within a quarter of the device memory
    while (grid > 1 && (size_t)grid * slab > prop.totalGlobalMem / 4) grid >>= 1;
    Buffers buf;
    pw_cell_in d_in = *in;
    pw_cell_out d_out = *out;
    double *d_xyz, *d_lat = nullptr, *d_inv = nullptr, *d_cov, *d_mass, *d_oxyz;
    unsigned char *d_term, *d_slabs;
    int *d_nmol, *d_status, *d_off, *d_src;
    signed char* d_img;
    unsigned long long* d_counter;
    RB_TRY(buf.alloc(&d_xyz, sizeof(double) * 3 * n * F));
    RB_TRY(buf.alloc(&d_cov, sizeof(double) * n));
    RB_TRY(buf.alloc(&d_mass, sizeof(double) * n));
    RB_TRY(buf.alloc(&d_term, n));
    if (in->lattice) {
        RB_TRY(buf.alloc(&d_lat, sizeof(double) * 9 * F));
        RB_TRY(buf.alloc(&d_inv, sizeof(double) * 9 * F));
    }
    RB_TRY(buf.alloc(&d_nmol, sizeof(int) * F));
    RB_TRY(buf.alloc(&d_status, sizeof(int) * F));
    RB_TRY(buf.alloc(&d_off, sizeof(int) * F * (out->mols_cap + 1)));
    RB_TRY(buf.alloc(&d_src, sizeof(int) * F * out->atoms_cap));
    RB_TRY(buf.alloc(&d_img, (size_t)F * out->atoms_cap));
    RB_TRY(buf.alloc(&d_oxyz, sizeof(double) * 3 * F * out->atoms_cap));
    RB_TRY(buf.alloc(&d_slabs, (size_t)grid * slab));
    RB_TRY(buf.alloc(&d_counter, sizeof(unsigned long long)));
    RB_TRY(hipMemcpyAsync(d_xyz, in->xyz, sizeof(double) * 3 * n * F, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(d_cov, in->cov, sizeof(double) * n, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(d_mass, in->mass, sizeof(double) * n, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(d_term, in->terminal, n, hipMemcpyHostToDevice, st));
    if (in->lattice) {
        RB_TRY(hipMemcpyAsync(d_lat, in->lattice, sizeof(double) * 9 * F, hipMemcpyHostToDevice, st));
        RB_TRY(hipMemcpyAsync(d_inv, in->lattice_inv, sizeof(double) * 9 * F, hipMemcpyHostToDevice, st));
    }
    RB_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), st));
    RB_TRY(hipMemsetAsync(d_off, 0, sizeof(int) * F * (out->mols_cap + 1), st));
    d_in.xyz = d_xyz; d_in.lattice = d_lat; d_in.lattice_inv = d_inv;
    d_in.cov = d_cov; d_in.mass = d_mass; d_in.terminal = d_term;
    d_out.n_mol = d_nmol; d_out.status = d_status; d_out.mol_offset = d_off;
    d_out.src_atom = d_src; d_out.src_image = (int8_t*)d_img; d_out.xyz = d_oxyz;
    hipLaunchKernelGGL(pw_rebuild_kernel, dim3((unsigned)grid), dim3(RB_WAVES * 64), 0, st, d_in, d_out,
                       d_slabs, slab, d_counter);
    RB_TRY(hipGetLastError());
    RB_TRY(hipMemcpyAsync(out->n_mol, d_nmol, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->status, d_status, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->mol_offset, d_off, sizeof(int) * F * (out->mols_cap + 1), hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->src_atom, d_src, sizeof(int) * F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->src_image, d_img, (size_t)F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->xyz, d_oxyz, sizeof(double) * 3 * F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipStreamSynchronize(st));
    return PW_OK;
}
