// pw_common.hpp -- shared host/device definitions for the pore-geometry engine.
//
// The numerical core (pw_blas/pw_ext64/pw_lbfgsb/pw_simplex/pw_math/pw_unit) is
// single-source: hipcc compiles it for gfx950 inside pw_kernels.hip, and the
// CPU-only test harness tests/hostsim compiles the very same headers with g++
// so the control flow can be checked against the golden vectors in a container
// without a GPU.  Everything is FP64 and must be built with -ffp-contract=off:
// fused multiply-adds appear only where written (pw_fma).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PW_HD __host__ __device__
#define PW_D __device__
#define PW_NOINLINE __attribute__((noinline))
#else
#define PW_NOINLINE
#define PW_HD
#define PW_D
#endif

namespace pw {

PW_HD inline double pw_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PW_HD inline double pw_sqrt(double a) { return __builtin_sqrt(a); }
PW_HD inline double pw_abs(double a) { return __builtin_fabs(a); }
PW_HD inline double pw_max(double a, double b) { return a > b ? a : b; }   // no NaNs on this path
PW_HD inline double pw_min(double a, double b) { return a < b ? a : b; }

constexpr double PW_INF = __builtin_huge_val();

}  // namespace pw
