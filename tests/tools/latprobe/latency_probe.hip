// latency_probe.hip -- dependent-instruction latencies of ONE wave alone on its SIMD (gfx950): what a serial optimiser
// chain pays per operation.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off latency_probe.hip -o latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP 512
template <class F>
__device__ long long timed(F f) {
    long long t0 = clock64();
    f();
    long long t1 = clock64();
    return t1 - t0;
}
__global__ void probe(double* io, long long* out, int* idx) {
    __shared__ double lds[1024];
    __shared__ int ldsi[1024];
    const int lane = threadIdx.x;
    lds[lane] = io[lane];
    ldsi[lane] = idx[lane];
    __syncthreads();
    double a = io[0], b = io[1], c = io[2];
    long long t;
    int k = 0;
    // 0: empty timer pair
    t = timed([&] { asm volatile("" ::: "memory"); });
    if (lane == 0) out[k] = t; k++;
    // 1: dependent v_fma_f64
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) a = __builtin_fma(a, b, c);
    });
    if (lane == 0) out[k] = t; k++;
    // 2: dependent v_add_f64
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) a = a + b;
    });
    if (lane == 0) out[k] = t; k++;
    // 3: dependent division a = c / a
    t = timed([&] {
#pragma unroll 8
        for (int i = 0; i < REP; ++i) a = c / a;
    });
    if (lane == 0) out[k] = t; k++;
    // 4: dependent sqrt
    t = timed([&] {
#pragma unroll 8
        for (int i = 0; i < REP; ++i) a = __builtin_sqrt(a + 3.0);
    });
    if (lane == 0) out[k] = t; k++;
    // 5: dependent v_rcp_f64
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) a = __builtin_amdgcn_rcp(a);
    });
    if (lane == 0) out[k] = t; k++;
    // 6: dependent LDS round trip (pointer chase through ints)
    int p = lane;
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) p = ldsi[p];
    });
    if (lane == 0) out[k] = t; k++;
    // 7: dependent LDS double read + add (value feeds address)
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) { a = a + lds[(p + (int)a) & 1023]; }
    });
    if (lane == 0) out[k] = t; k++;
    // 8: dependent v_cndmask pair (select on double)
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) a = (a > b) ? c : a + 1.0;
    });
    if (lane == 0) out[k] = t; k++;
    // 9: readlane broadcast of a double + add
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            union { double d; int w[2]; } u;
            u.d = a;
            u.w[0] = __builtin_amdgcn_readlane(u.w[0], 5);
            u.w[1] = __builtin_amdgcn_readlane(u.w[1], 5);
            a = u.d + b;
        }
    });
    if (lane == 0) out[k] = t; k++;
    // 10: divergent-form branch on a uniform value (exec masking), dependent
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            if (a > c) a = a - b; else a = a + b * 0.5;
        }
    });
    if (lane == 0) out[k] = t; k++;
    // 11: the same with a wave-uniform branch condition (ballot)
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            if (__builtin_amdgcn_ballot_w64(a > c) != 0) a = a - b; else a = a + b * 0.5;
        }
    });
    if (lane == 0) out[k] = t; k++;
    // 12: four INDEPENDENT fma chains (throughput of one wave)
    {
        double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3;
        t = timed([&] {
#pragma unroll 16
            for (int i = 0; i < REP; ++i) {
                a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
            }
        });
        a = a0 + a1 + a2 + a3;
    }
    if (lane == 0) out[k] = t; k++;
    // 13: DPP row_shr min step (dependent)
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) {
            union { double d; int w[2]; } u, v;
            u.d = a;
            v.w[0] = __builtin_amdgcn_update_dpp(u.w[0], u.w[0], 0x111, 0xf, 0xf, false);
            v.w[1] = __builtin_amdgcn_update_dpp(u.w[1], u.w[1], 0x111, 0xf, 0xf, false);
            a = __builtin_fmin(a, v.d) + 1.0;
        }
    });
    if (lane == 0) out[k] = t; k++;
    // 14: scratch (private array, dynamic index) round trip
    {
        double priv[16];
        for (int i = 0; i < 16; ++i) priv[i] = a + i;
        int q = p & 15;
        t = timed([&] {
#pragma unroll 8
            for (int i = 0; i < REP; ++i) { priv[q] = a; q = (q + 5) & 15; a = a + priv[q]; }
        });
    }
    if (lane == 0) out[k] = t; k++;
    // 15: dependent integer VALU add
    t = timed([&] {
#pragma unroll 16
        for (int i = 0; i < REP; ++i) p = p * 3 + 1;
    });
    if (lane == 0) out[k] = t; k++;
    // 16: s_memtime pair + atomicAdd (what one fine timer costs)
    t = timed([&] {
        for (int i = 0; i < 64; ++i) {
            long long s0 = clock64();
            a = a + b;
            long long s1 = clock64();
            if (lane == 0) atomicAdd((unsigned long long*)&out[40], (unsigned long long)(s1 - s0));
        }
    });
    if (lane == 0) out[k] = t * (REP / 64); k++;
    io[lane] = a + p;
}
int main() {
    double h[64];
    for (int i = 0; i < 64; ++i) h[i] = 1.0 + 1e-3 * i;
    int hi[1024];
    for (int i = 0; i < 1024; ++i) hi[i] = (i * 7 + 3) & 1023;
    double* d; long long* o; int* di;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 8); hipMalloc(&di, sizeof(hi));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemcpy(di, hi, sizeof(hi), hipMemcpyHostToDevice);
    hipMemset(o, 0, 64 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe, 1, 64, 0, 0, d, o, di); hipDeviceSynchronize(); }
    long long ho[64];
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* names[] = {"empty timer pair (total cycles)", "v_fma_f64 dependent", "v_add_f64 dependent", "a = c / a", "sqrt(a + 3)",
                           "v_rcp_f64 dependent", "LDS int pointer chase", "LDS double read + add + index", "select on double + add",
                           "readlane double + add", "exec-mask branch (uniform data)", "ballot-uniform branch", "4 independent fma (per group)",
                           "DPP row_shr min + add", "scratch store + load (dynamic index)", "v_mul_lo + add int", "fine timer pair + atomic"};
    printf("%-40s %10s\n", "operation", "cycles/op");
    for (int k = 0; k < 17; ++k) printf("%-40s %10.1f\n", names[k], k == 0 ? (double)ho[k] : (double)(ho[k] - ho[0]) / REP);
    return 0;
}
