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
