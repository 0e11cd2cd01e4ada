// How many HIP streams of one process run kernels AT THE SAME TIME on this device?  (Round 6: the pipeline keeps one
// stream pair per analysis in flight; the question is whether more than 2 + 2 x 4 streams get hardware queues of their own
// when GPU_MAX_HW_QUEUES asks for them.)   usage:  GPU_MAX_HW_QUEUES=32 ./stream_probe 12 16 20 24 32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void probe(int* started, int expected, int* together) {
    if (threadIdx.x != 0) return;
    atomicAdd(started, 1);
    long long t0 = wall_clock64();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > 2000000ll) return;     // 20 ms
    }
    atomicAdd(together, 1);
}
int main(int argc, char** argv) {
    int* flags = nullptr;
    if (hipMalloc(&flags, 8) != hipSuccess) { printf("no device\n"); return 1; }
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    for (int a = 1; a < argc; ++a) {
        const int k = atoi(argv[a]);
        std::vector<hipStream_t> st(k);
        for (int i = 0; i < k; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        int best = 0;
        for (int round = 0; round < 4; ++round) {
            hipMemset(flags, 0, 8);
            hipDeviceSynchronize();
            for (int i = 0; i < k; ++i) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, st[i], flags, round == 0 ? 0 : k, flags + 1);
            hipDeviceSynchronize();
            int got[2];
            hipMemcpy(got, flags, 8, hipMemcpyDeviceToHost);
            if (round > 0 && got[1] > best) best = got[1];
        }
        printf("GPU_MAX_HW_QUEUES=%s streams %d: %d ran together\n", q ? q : "unset", k, best);
        for (int i = 0; i < k; ++i) hipStreamDestroy(st[i]);
    }
    return 0;
}
