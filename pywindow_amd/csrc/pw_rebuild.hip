// pw_rebuild.hip -- gfx950 kernel and C ABI entry of the periodic pre-processing
// (include/pywindow_amd.h: pw_discrete_molecules; reference utilities.py:768-1085).
//
// One persistent workgroup (4 wavefronts) re-assembles one frame at a time; frames are
// handed out by an atomic counter.  All state of a frame lives in a per-team slab of
// global memory that stays L2-resident (value coordinates of the 27 images, candidate
// lists, visit stamps); the traversal itself is a chain of short dependent steps, so the
// launch is sized for many concurrent frames rather than for wide teams.
#include <hip/hip_runtime.h>
#include <time.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/pywindow_amd.h"
#include "pw_host.hpp"
#include "pw_rebuild.hpp"
#include "pw_team.hpp"

using namespace pw;

extern "C" char* pw_internal_error_buffer(void);   // pw_kernels.hip (512 bytes, thread local)
extern "C" int pw_context_device(pw_context* ctx);
extern "C" int pw_internal_pool(pw_context* ctx, size_t bytes, void** out);   // pw_kernels.hip
extern "C" int pw_internal_block_take(pw_context* ctx, size_t bytes, void** out, size_t* got);
extern "C" void pw_internal_block_give(pw_context* ctx, void* p, size_t bytes);

namespace {

constexpr int RB_WAVES = 4;

__global__ void __launch_bounds__(RB_WAVES * 64)
pw_rebuild_kernel(pw_cell_in in, pw_cell_out out, unsigned char* __restrict__ slabs, size_t slab_bytes,
                  unsigned long long* counter, int with_bits, int with_scan) {
    using T = DeviceTeam<RB_WAVES>;
    extern __shared__ __attribute__((aligned(16))) unsigned char fast[];   // RebuildWs::fast_bytes
    __shared__ long s_frame;
    const int n = in.n_atoms;
    RebuildWs* w = RebuildWs::carve(slabs + (size_t)blockIdx.x * slab_bytes, n, in.rebuild, T::SIZE);
    w->attach_fast(fast, n, in.rebuild, with_bits != 0, with_scan != 0);
    // (the walk is one wave's chain of dependent look-ups: beside the bulk waves of an analysis it must win the issue
    // arbitration, like the optimiser chains -- the next piece's analysis waits for this launch)
    __builtin_amdgcn_s_setprio(2);
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

// temporaries of one call, taken from the context's cache of device blocks and given back at the end: no
// hipMalloc / hipFree -- and no device-wide wait -- per call once the cache is warm.  A block may only go
// back once nothing queued on the stream can still touch it: the success paths end with a stream
// synchronisation and say so (`synced`); every other way out (an error return half-way) waits here first.
struct Buffers {
    static constexpr int CAP = 32;
    pw_context* ctx = nullptr;
    hipStream_t st = nullptr;
    bool synced = false;
    void* p[CAP];
    size_t bytes_[CAP];
    int n = 0;
    ~Buffers() {
        if (n > 0 && !synced) (void)hipStreamSynchronize(st);
        for (int i = 0; i < n; ++i) if (p[i]) pw_internal_block_give(ctx, p[i], bytes_[i]);
    }
    template <class X> hipError_t alloc(X** out, size_t bytes) {
        if (n >= CAP || !ctx) return hipErrorOutOfMemory;
        void* q = nullptr;
        size_t got = 0;
        if (pw_internal_block_take(ctx, bytes ? bytes : 8, &q, &got) != PW_OK) return hipErrorOutOfMemory;
        *out = (X*)q;
        p[n] = q; bytes_[n] = got; ++n;
        return hipSuccess;
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

extern "C" void* pw_internal_rebuild_stream(pw_context* ctx);      // (pw_kernels.hip: highest priority, see there)
extern "C" int pw_internal_resident_adopt(pw_context* ctx, long n_units, long n_atoms, int nmax, long* d_offset,
                                          double* d_xyz, double* d_vdw, double* d_mass, const size_t* part_bytes,
                                          pw_resident** out);

namespace {

// device-side result of the rebuild launch (freed with the Buffers object that owns it)
struct DeviceCells {
    int *n_mol, *status, *off, *src;
    signed char* img;
    double *oxyz, *mass;
};

bool args_ok(pw_context* ctx, const pw_cell_in* in) {
    return ctx && in && in->n_frames >= 0 && in->n_atoms > 0 && in->n_atoms < (1 << RB_NB_IMG_SHIFT) && in->xyz && in->cov && in->mass && in->terminal;
}

// upload the frames, run the rebuild kernel; outputs stay on the device
int rebuild_on_device(pw_context* ctx, const pw_cell_in* in, int atoms_cap, int mols_cap, Buffers& buf,
                      DeviceCells* dev) {
    if (in->rebuild && (!in->lattice || !in->lattice_inv)) {
        snprintf(pw_internal_error_buffer(), 512, "rebuild needs the lattice and its inverse");
        return PW_E_BAD_ARG;
    }
    if (in->lattice && !in->lattice_inv) return PW_E_BAD_ARG;
    // (the callers -- the two entry points below -- have made the context's device current)
    hipStream_t st = (hipStream_t)pw_internal_rebuild_stream(ctx);
    const long F = (long)in->n_frames;
    const int n = in->n_atoms;
    hipDeviceProp_t prop;
    RB_TRY(hipGetDeviceProperties(&prop, pw_context_device(ctx)));
    size_t slab = (RebuildWs::bytes(n, in->rebuild, RB_WAVES * 64) + 255) & ~(size_t)255;
    int per_cu = 4;
    if (const char* e = getenv("PW_RB_TEAMS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) per_cu = v; }
    long grid = (long)prop.multiProcessorCount * per_cu;
    if (grid > F) grid = F;
    // keep the slabs within a quarter of the device memory
    while (grid > 1 && (size_t)grid * slab > prop.totalGlobalMem / 4) grid >>= 1;
    pw_cell_in d_in = *in;
    pw_cell_out d_out;
    double *d_xyz, *d_lat = nullptr, *d_inv = nullptr, *d_cov;
    unsigned char *d_term, *d_slabs;
    unsigned long long* d_counter;
    RB_TRY(buf.alloc(&d_xyz, sizeof(double) * 3 * n * F));
    RB_TRY(buf.alloc(&d_cov, sizeof(double) * n));
    RB_TRY(buf.alloc(&dev->mass, sizeof(double) * n));
    RB_TRY(buf.alloc(&d_term, n));
    if (in->lattice) {
        RB_TRY(buf.alloc(&d_lat, sizeof(double) * 9 * F));
        RB_TRY(buf.alloc(&d_inv, sizeof(double) * 9 * F));
    }
    RB_TRY(buf.alloc(&dev->n_mol, sizeof(int) * F));
    RB_TRY(buf.alloc(&dev->status, sizeof(int) * F));
    RB_TRY(buf.alloc(&dev->off, sizeof(int) * F * (mols_cap + 1)));
    RB_TRY(buf.alloc(&dev->src, sizeof(int) * F * atoms_cap));
    RB_TRY(buf.alloc(&dev->img, (size_t)F * atoms_cap));
    RB_TRY(buf.alloc(&dev->oxyz, sizeof(double) * 3 * F * atoms_cap));
    {
        // the team slabs come from the context's pool (kept between calls)
        void* pool = nullptr;
        int rcp = pw_internal_pool(ctx, (size_t)grid * slab, &pool);
        if (rcp != PW_OK) return rcp;
        d_slabs = (unsigned char*)pool;
    }
    RB_TRY(buf.alloc(&d_counter, sizeof(unsigned long long)));
    RB_TRY(hipMemcpyAsync(d_xyz, in->xyz, sizeof(double) * 3 * n * F, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(d_cov, in->cov, sizeof(double) * n, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(dev->mass, in->mass, sizeof(double) * n, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemcpyAsync(d_term, in->terminal, n, hipMemcpyHostToDevice, st));
    if (in->lattice) {
        RB_TRY(hipMemcpyAsync(d_lat, in->lattice, sizeof(double) * 9 * F, hipMemcpyHostToDevice, st));
        RB_TRY(hipMemcpyAsync(d_inv, in->lattice_inv, sizeof(double) * 9 * F, hipMemcpyHostToDevice, st));
    }
    RB_TRY(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), st));
    RB_TRY(hipMemsetAsync(dev->off, 0, sizeof(int) * F * (mols_cap + 1), st));
    d_in.xyz = d_xyz; d_in.lattice = d_lat; d_in.lattice_inv = d_inv;
    d_in.cov = d_cov; d_in.mass = dev->mass; d_in.terminal = d_term;
    d_out.atoms_cap = atoms_cap; d_out.mols_cap = mols_cap;
    d_out.n_mol = dev->n_mol; d_out.status = dev->status; d_out.mol_offset = dev->off;
    d_out.src_atom = dev->src; d_out.src_image = (int8_t*)dev->img; d_out.xyz = dev->oxyz;
    // team-shared memory: the hit segments, and the two visit bit sets when they fit beside them
    int with_bits = RebuildWs::fast_bytes(n, in->rebuild, true) <= 96 * 1024 ? 1 : 0;
    // ... and the single-precision coordinates of the candidate scan while two teams still fit a CU
    int with_scan = RebuildWs::fast_bytes(n, in->rebuild, with_bits != 0, true) <= 79 * 1024 ? 1 : 0;
    size_t lds = RebuildWs::fast_bytes(n, in->rebuild, with_bits != 0, with_scan != 0);
    {
        // the limit, not a request: set once per device (host threads may launch concurrently)
        static std::atomic<unsigned long long> done{0};
        unsigned long long bit = 1ull << (pw_context_device(ctx) & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            RB_TRY(hipFuncSetAttribute((const void*)pw_rebuild_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       128 * 1024));
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(pw_rebuild_kernel, dim3((unsigned)grid), dim3(RB_WAVES * 64), lds, st, d_in, d_out,
                       d_slabs, slab, d_counter, with_bits, with_scan);
    RB_TRY(hipGetLastError());
    return PW_OK;
}

// exclusive scans over the frames: first unit and first atom of every frame (F + 1 entries each)
__global__ void __launch_bounds__(256)
rb_scan_kernel(long F, int mols_cap, const int* __restrict__ n_mol, const int* __restrict__ off,
               long* __restrict__ unit_base, long* __restrict__ atom_base) {
    // one block: a thread sums a run of consecutive frames, the runs are scanned through team-shared memory
    // (a single thread walking the frames paid two dependent memory round trips per frame)
    __shared__ long su[256], sa[256];
    const int t = threadIdx.x;
    const long per = (F + 255) / 256;
    const long f0 = t * per, f1 = f0 + per < F ? f0 + per : F;
    long u = 0, a = 0;
    for (long f = f0; f < f1; ++f) {
        const int m = n_mol[f];
        u += m;
        a += off[f * (mols_cap + 1) + m];
    }
    su[t] = u; sa[t] = a;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        long xu = 0, xa = 0;
        if (t >= d) { xu = su[t - d]; xa = sa[t - d]; }
        __syncthreads();
        su[t] += xu; sa[t] += xa;
        __syncthreads();
    }
    u = su[t] - u; a = sa[t] - a;              // exclusive
    for (long f = f0; f < f1; ++f) {
        unit_base[f] = u;
        atom_base[f] = a;
        const int m = n_mol[f];
        u += m;
        a += off[f * (mols_cap + 1) + m];
    }
    if (t == 255) { unit_base[F] = su[255]; atom_base[F] = sa[255]; }
}

// the molecules of all frames as one ragged batch: offsets, coordinates, radii and masses by source atom
__global__ void rb_gather_kernel(long F, int atoms_cap, int mols_cap, const int* __restrict__ n_mol,
                                 const int* __restrict__ off, const int* __restrict__ src,
                                 const double* __restrict__ oxyz, const double* __restrict__ vdw,
                                 const double* __restrict__ mass, const long* __restrict__ unit_base,
                                 const long* __restrict__ atom_base, long* __restrict__ d_offset,
                                 double* __restrict__ xyz, double* __restrict__ uv, double* __restrict__ um,
                                 int* __restrict__ nmax) {
    const long f = blockIdx.x;
    const int m = n_mol[f];
    const int* fo = off + f * (mols_cap + 1);
    const long ub = unit_base[f], ab = atom_base[f];
    for (int k = threadIdx.x; k < m; k += blockDim.x) {
        d_offset[ub + k] = ab + fo[k];
        atomicMax(nmax, fo[k + 1] - fo[k]);
    }
    if (f == F - 1 && threadIdx.x == 0) d_offset[ub + m] = ab + fo[m];
    const int na = fo[m];
    const int* fs = src + f * atoms_cap;
    const double* fx = oxyz + 3 * f * atoms_cap;
    for (int j = threadIdx.x; j < na; j += blockDim.x) {
        long d = ab + j;
        xyz[3 * d] = fx[3 * j]; xyz[3 * d + 1] = fx[3 * j + 1]; xyz[3 * d + 2] = fx[3 * j + 2];
        int q = fs[j];
        uv[d] = vdw[q];
        um[d] = mass[q];
    }
}

}  // namespace

extern "C" int pw_discrete_molecules(pw_context* ctx, const pw_cell_in* in, const pw_cell_out* out) {
    if (!args_ok(ctx, in) || !out || !out->n_mol || !out->status || !out->mol_offset || !out->src_atom ||
        !out->src_image || !out->xyz || out->atoms_cap <= 0 || out->mols_cap <= 0)
        return PW_E_BAD_ARG;
    if (in->n_frames == 0) return PW_OK;
    PW_LOCK_CONTEXT(ctx);
    DeviceScope dev_scope_;
    if (pw_context_device(ctx) < 0) {
        snprintf(pw_internal_error_buffer(), 512, "not part of the host path (device = -1 runs the analysis only)");
        return PW_E_NO_DEVICE;
    }
    RB_TRY(dev_scope_.enter(pw_context_device(ctx)));
    Buffers buf;
    buf.ctx = ctx;
    buf.st = (hipStream_t)pw_internal_rebuild_stream(ctx);
    DeviceCells dev;
    int rc = rebuild_on_device(ctx, in, out->atoms_cap, out->mols_cap, buf, &dev);
    if (rc != PW_OK) return rc;
    hipStream_t st = (hipStream_t)pw_internal_rebuild_stream(ctx);
    const long F = (long)in->n_frames;
    RB_TRY(hipMemcpyAsync(out->n_mol, dev.n_mol, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->status, dev.status, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->mol_offset, dev.off, sizeof(int) * F * (out->mols_cap + 1), hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->src_atom, dev.src, sizeof(int) * F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->src_image, dev.img, (size_t)F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(out->xyz, dev.oxyz, sizeof(double) * 3 * F * out->atoms_cap, hipMemcpyDeviceToHost, st));
    RB_TRY(hipStreamSynchronize(st));
    buf.synced = true;
    return PW_OK;
}

extern "C" int pw_resident_from_cells(pw_context* ctx, const pw_cell_in* in, const double* vdw, int32_t atoms_cap,
                                      int32_t mols_cap, pw_resident** res, int32_t* n_mol, int32_t* status) {
    if (!args_ok(ctx, in) || !vdw || !res || !n_mol || !status || atoms_cap <= 0 || mols_cap <= 0 || in->n_frames <= 0)
        return PW_E_BAD_ARG;
    *res = nullptr;
    PW_LOCK_CONTEXT(ctx);
    DeviceScope dev_scope_;
    if (pw_context_device(ctx) < 0) {
        snprintf(pw_internal_error_buffer(), 512, "not part of the host path (device = -1 runs the analysis only)");
        return PW_E_NO_DEVICE;
    }
    RB_TRY(dev_scope_.enter(pw_context_device(ctx)));
    // PW_RB_TIMING=1: the host-side legs of this call on stderr (ms): queued = uploads and the re-assembly launch queued,
    // rebuilt = that launch and the scan finished (first wait), gathered = the ragged batch built (second wait), adopted
    const bool rb_timing = getenv("PW_RB_TIMING") && getenv("PW_RB_TIMING")[0] == '1';
    timespec rt0;
    clock_gettime(CLOCK_MONOTONIC, &rt0);
    double rt[4] = {0, 0, 0, 0};
    auto rt_mark = [&](int k) {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        rt[k] = (t.tv_sec - rt0.tv_sec) * 1e3 + (t.tv_nsec - rt0.tv_nsec) * 1e-6;
    };
    Buffers buf;
    buf.ctx = ctx;
    buf.st = (hipStream_t)pw_internal_rebuild_stream(ctx);
    DeviceCells dev;
    int rc = rebuild_on_device(ctx, in, atoms_cap, mols_cap, buf, &dev);
    if (rc != PW_OK) return rc;
    rt_mark(0);
    hipStream_t st = (hipStream_t)pw_internal_rebuild_stream(ctx);
    const long F = (long)in->n_frames;
    const int n = in->n_atoms;
    long *d_ubase, *d_abase;
    double* d_vdw_atom;
    int* d_nmax;
    RB_TRY(buf.alloc(&d_ubase, sizeof(long) * (F + 1)));
    RB_TRY(buf.alloc(&d_abase, sizeof(long) * (F + 1)));
    RB_TRY(buf.alloc(&d_vdw_atom, sizeof(double) * n));
    RB_TRY(buf.alloc(&d_nmax, sizeof(int)));
    RB_TRY(hipMemcpyAsync(d_vdw_atom, vdw, sizeof(double) * n, hipMemcpyHostToDevice, st));
    RB_TRY(hipMemsetAsync(d_nmax, 0, sizeof(int), st));
    hipLaunchKernelGGL(rb_scan_kernel, dim3(1), dim3(256), 0, st, F, (int)mols_cap, dev.n_mol, dev.off, d_ubase, d_abase);
    RB_TRY(hipGetLastError());
    long totals[2] = {0, 0};
    RB_TRY(hipMemcpyAsync(n_mol, dev.n_mol, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(status, dev.status, sizeof(int) * F, hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(&totals[0], d_ubase + F, sizeof(long), hipMemcpyDeviceToHost, st));
    RB_TRY(hipMemcpyAsync(&totals[1], d_abase + F, sizeof(long), hipMemcpyDeviceToHost, st));
    RB_TRY(hipStreamSynchronize(st));
    rt_mark(1);
    buf.synced = true;             // (until the gather below is queued)
    for (long f = 0; f < F; ++f)
        if (status[f] & (PW_RB_ATOMS_OVERFLOW | PW_RB_MOLS_OVERFLOW)) {
            snprintf(pw_internal_error_buffer(), 512, "frame %ld needs more than %d atoms / %d molecules", f,
                     atoms_cap, mols_cap);
            return PW_E_TOO_LARGE;
        }
    const long U = totals[0], A = totals[1];
    if (U == 0) return PW_OK;      // nothing to analyse: *res stays NULL
    long* d_offset = nullptr;
    double *d_xyz = nullptr, *d_uv = nullptr, *d_um = nullptr;
    size_t part_bytes[4] = {0, 0, 0, 0};
    auto drop = [&]() {
        (void)hipStreamSynchronize(st);          // (the gather may be running on these blocks)
        pw_internal_block_give(ctx, d_offset, part_bytes[0]);
        pw_internal_block_give(ctx, d_xyz, part_bytes[1]);
        pw_internal_block_give(ctx, d_uv, part_bytes[2]);
        pw_internal_block_give(ctx, d_um, part_bytes[3]);
    };
    auto take = [&](void** q, size_t bytes, int k) { return pw_internal_block_take(ctx, bytes, q, &part_bytes[k]) == PW_OK ? hipSuccess : hipErrorOutOfMemory; };
#define RBF_TRY(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(pw_internal_error_buffer(), 512, "%s: %s", #call, hipGetErrorString(e_)); \
            drop();                                                                        \
            return PW_E_HIP;                                                               \
        }                                                                                  \
    } while (0)
    RBF_TRY(take((void**)&d_offset, sizeof(long) * (U + 1), 0));
    RBF_TRY(take((void**)&d_xyz, sizeof(double) * 3 * A, 1));
    RBF_TRY(take((void**)&d_uv, sizeof(double) * A, 2));
    RBF_TRY(take((void**)&d_um, sizeof(double) * A, 3));
    buf.synced = false;
    hipLaunchKernelGGL(rb_gather_kernel, dim3((unsigned)F), dim3(256), 0, st, F, (int)atoms_cap, (int)mols_cap,
                       dev.n_mol, dev.off, dev.src, dev.oxyz, d_vdw_atom, dev.mass, d_ubase, d_abase, d_offset, d_xyz,
                       d_uv, d_um, d_nmax);
    RBF_TRY(hipGetLastError());
    int nmax = 0;
    RBF_TRY(hipMemcpyAsync(&nmax, d_nmax, sizeof(int), hipMemcpyDeviceToHost, st));
    RBF_TRY(hipStreamSynchronize(st));
    rt_mark(2);
    buf.synced = true;
#undef RBF_TRY
    rc = pw_internal_resident_adopt(ctx, U, A, nmax, d_offset, d_xyz, d_uv, d_um, part_bytes, res);
    if (rc != PW_OK) drop();
    rt_mark(3);
    if (rb_timing)
        fprintf(stderr, "pywindow_amd: pw_resident_from_cells %ld frames: queued %.2f rebuilt %.2f gathered %.2f adopted %.2f ms\n", F,
                rt[0], rt[1] - rt[0], rt[2] - rt[1], rt[3] - rt[2]);
    return rc;
}
