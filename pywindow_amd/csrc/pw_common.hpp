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

// ---- a division whose divisor is known in advance -------------------------------------------------------------
// On gfx950 a / b is eleven dependent instructions: v_div_scale x2, v_rcp_f64, two Newton steps on the reciprocal,
// q = a * r, rem = fma(-b, q, a), v_div_fmas (= fma(rem, r, q)), v_div_fixup.  Everything up to the refined
// reciprocal r depends on b alone, and the two scale instructions, div_fmas' scaling and the fix-up do NOTHING while
// both exponents are unremarkable (ISA: V_DIV_SCALE_F64 scales for a denormal or huge divisor, an exponent
// difference >= 768, a quotient that would be denormal, a dividend exponent <= 53).  pw_recip_hw(b) is that r --
// the same instructions in the same order -- and pw_div_r(a, b, r) the remaining three, so the quotient has the
// bits of a / b (checked on the device over 2^26 operand pairs, tests/test_gpu_api.py); operands outside the plain
// range, a zero divisor included, take a / b itself.  A solve that divides by the same diagonal at every step
// (pw_lbfgsb.hpp) pays three dependent instructions per step instead of eleven.  On the host both are the plain
// division (IEEE: the same bits).
PW_HD inline bool pw_plain_exponent(double a) {
    // biased exponent in [700, 1400]: 2^-323 <= |a| < 2^378
    union { double d; unsigned long long u; } c;
    c.d = a;
    const unsigned h = (unsigned)(c.u >> 32) & 0x7ff00000u;
    return (h - 0x2bc00000u) <= 0x2bc00000u;
}
PW_HD inline double pw_recip_hw(double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!pw_plain_exponent(b)) return 0.0;          // pw_div_r divides for real
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e0 = __builtin_fma(-b, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-b, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
#else
    (void)b;
    return 1.0;
#endif
}
// the three instructions alone: for callers that check afterwards that every dividend was plain and non-zero (and
// repeat the computation with pw_div_r if one was not)
PW_HD inline double pw_div_ru(double a, double b, double r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
#else
    (void)r;
    return a / b;
#endif
}
PW_HD inline double pw_div_r(double a, double b, double r) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double q = a * r;
    const double rem = __builtin_fma(-b, q, a);
    double res = __builtin_fma(rem, r, q);
    const bool zero = a == 0.0;
    res = zero ? q : res;                          // (+-0 / b: the sign of the product)
    const bool fast = (pw_plain_exponent(a) || zero) && r != 0.0;
    if (__builtin_expect(!fast, 0)) res = a / b;
    return res;
#else
    (void)r;
    return a / b;
#endif
}

}  // namespace pw
