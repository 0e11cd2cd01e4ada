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
#include <type_traits>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PW_HD __host__ __device__
#define PW_D __device__
#ifdef PW_STATIC_NOINLINE
#define PW_NOINLINE __attribute__((noinline)) static
#else
#define PW_NOINLINE __attribute__((noinline))
#endif
#else
#define PW_NOINLINE
#define PW_HD
#define PW_D
#endif

// LDS (address space 3) qualification: pointers that always point into the team's
// shared memory carry it on the device so that loads/stores become ds_* instead of
// flat_* instructions; PW_ASSUME_LDS tells the optimiser the same about `this`.
// PW_GENERIC_TEAM_MEM (pw_kernels_big.hip): the team's shared block lives in GLOBAL memory -- molecules whose
// coordinates do not fit the 160 KB of a CU's LDS -- so nothing is qualified and nothing is assumed.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PW_GENERIC_TEAM_MEM)
#define PW_LDS __attribute__((address_space(3)))
#define PW_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#define PW_IS_LDS(p) __builtin_amdgcn_is_shared((const void*)(p))
#else
#define PW_LDS
#define PW_ASSUME_LDS(p) do {} while (0)
#define PW_IS_LDS(p) true
#endif
// Arrays carved from the idle part of the team's LDS fall back to the global workspace when they do not
// fit, so their pointers are generic and every access a flat_* instruction (issued through the vector
// memory path, waited for on both counters).  Hot loops over such arrays are written once as a generic
// lambda over the pointer types and entered through PW_WITH_LDS*: when the arrays are in LDS -- the
// usual case -- the loop runs on address-space-3 pointers (ds_* instructions).
#define PW_AS_LDS(p) ((PW_LDS std::remove_pointer_t<decltype(p)>*)(p))

namespace pw {

typedef PW_LDS double ldouble;
typedef PW_LDS int lint;

PW_HD inline double pw_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// (-2 * g) + a in ONE instruction: the product by two is exact, so the fused form rounds once exactly
// where the two-instruction form of the distance primitive rounds -- the same bits, one VALU operation
// less in every atom-point distance
PW_HD inline double pw_m2add(double g, double a) { return __builtin_fma(-2.0, g, a); }
PW_HD inline double pw_sqrt(double a) { return __builtin_sqrt(a); }
PW_HD inline double pw_abs(double a) { return __builtin_fabs(a); }
PW_HD inline double pw_max(double a, double b) { return a > b ? a : b; }   // no NaNs on this path
PW_HD inline double pw_min(double a, double b) { return a < b ? a : b; }

constexpr double PW_INF = __builtin_huge_val();

}  // namespace pw
