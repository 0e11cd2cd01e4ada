// pw_host.hpp -- host-side helpers shared by the translation units of libpywindow_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace pw {

// Every C-ABI entry point works on its context's device and hands the calling thread back the
// device it came with: the HIP runtime (and its per-thread current device) is shared with the
// application -- with PyTorch in a one-process-per-GPU job -- and a library call must not move it.
struct DeviceScope {
    int prev = -1;
    bool moved = false;
    hipError_t enter(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) return hipSuccess;
        hipError_t e = hipSetDevice(device);
        moved = (e == hipSuccess);
        return e;
    }
    ~DeviceScope() {
        if (moved && prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace pw

// Threads (include/pywindow_amd.h, "Threads"): every entry point that takes a context holds the context's
// mutex for the duration of the call, so calls from several threads on ONE context are serialised and
// contexts never share mutable state.  Recursive: entry points are built from each other
// (pw_analysis_batch = upload + launch + download).
struct pw_context;
extern "C" void pw_internal_lock(pw_context* ctx);
extern "C" void pw_internal_unlock(pw_context* ctx);
struct PwContextLock {
    pw_context* c;
    explicit PwContextLock(pw_context* c_) : c(c_) { if (c) pw_internal_lock(c); }
    ~PwContextLock() { if (c) pw_internal_unlock(c); }
    PwContextLock(const PwContextLock&) = delete;
    PwContextLock& operator=(const PwContextLock&) = delete;
};
#define PW_LOCK_CONTEXT(c) PwContextLock ctx_lock_(c)
