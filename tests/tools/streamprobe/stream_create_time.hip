// What creating the pipeline's ten HIP streams costs in a fresh process, one after the other and from several threads
// (round 6: pw_context_create spends 140-190 ms of its 210-330 ms in "streams + events").
//   ./stream_create_time            sequential      ./stream_create_time 4      four threads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void nop() {}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 1;
    double t0 = now_ms();
    int n = 0;
    (void)hipGetDeviceCount(&n);
    double t1 = now_ms();
    (void)hipSetDevice(0);
    void* p = nullptr;
    (void)hipMalloc(&p, 1024);
    double t2 = now_ms();
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    std::vector<hipStream_t> st(10);
    std::vector<double> each(10);
    auto make = [&](int i) {
        double a = now_ms();
        (void)hipSetDevice(0);
        if (i < 4) (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
        else if (i == 4) (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, lo);
        else (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        each[i] = now_ms() - a;
    };
    if (threads <= 1) for (int i = 0; i < 10; ++i) make(i);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { for (int i = t; i < 10; i += threads) make(i); });
        for (auto& x : th) x.join();
    }
    double t3 = now_ms();
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st[i]);
    (void)hipDeviceSynchronize();
    double t4 = now_ms();
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st[i]);
    (void)hipDeviceSynchronize();
    double t5 = now_ms();
    printf("threads %d: hipGetDeviceCount %.1f ms | first hipMalloc %.1f | 10 streams %.1f (", threads, t1 - t0, t2 - t1, t3 - t2);
    for (int i = 0; i < 10; ++i) printf("%.1f ", each[i]);
    printf(") | first launch on each %.1f | second %.2f\n", t4 - t3, t5 - t4);
    return 0;
}
