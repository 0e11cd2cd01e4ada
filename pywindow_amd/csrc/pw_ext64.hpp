// pw_ext64.hpp -- software emulation of the x87 80-bit arithmetic OpenBLAS'
// dnrm2 kernel (kernel/x86_64/nrm2.S) performs: squares and the running sum are
// rounded to a 64-bit significand (round-to-nearest-even), fsqrt likewise, and
// the result is rounded once more to double when it is returned.  SciPy's
// L-BFGS-B port calls dnrm2 for |d| (line-search set-up) and for |y|^2 (the
// BFGS scaling theta), so the double rounding is visible in the iterates.
//
// Positive values only; no overflow/underflow/denormal handling (inputs here are
// O(1e-10 .. 1e3)).  Integer-only, so host and gfx950 agree bit for bit.
#pragma once
#include "pw_common.hpp"

namespace pw {

typedef unsigned __int128 u128;

struct Ext64 {
    uint64_t m;  // significand, MSB set unless zero
    int e;       // value = m * 2^e
};

// round a 128-bit significand (value = v * 2^e) to 64 bits, nearest-even
PW_HD inline Ext64 ext_round128(u128 v, int e) {
    Ext64 r;
    if (v == 0) { r.m = 0; r.e = 0; return r; }
    // position of the top bit
    uint64_t hi = (uint64_t)(v >> 64);
    int top;  // index of MSB in v
    if (hi) top = 127 - __builtin_clzll(hi);
    else top = 63 - __builtin_clzll((uint64_t)v);
    if (top <= 63) {
        r.m = (uint64_t)v << (63 - top);
        r.e = e - (63 - top);
        return r;
    }
    int sh = top - 63;  // bits to drop, 1..64
    u128 kept = v >> sh;
    u128 rem = v & ((((u128)1) << sh) - 1);
    u128 half = ((u128)1) << (sh - 1);
    uint64_t m = (uint64_t)kept;
    int ee = e + sh;
    if (rem > half || (rem == half && (m & 1))) {
        m += 1;
        if (m == 0) {  // carry out of 64 bits
            m = 0x8000000000000000ull;
            ee += 1;
        }
    }
    r.m = m;
    r.e = ee;
    return r;
}

// RN64(x*x) for a finite double x
PW_HD inline Ext64 ext_square(double x) {
    union { double d; uint64_t u; } c;
    c.d = x;
    uint64_t frac = c.u & 0x000fffffffffffffull;
    int be = (int)((c.u >> 52) & 0x7ff);
    Ext64 z; z.m = 0; z.e = 0;
    if (be == 0 && frac == 0) return z;
    uint64_t m53;
    int e;
    if (be == 0) { m53 = frac; e = -1074; }
    else { m53 = frac | 0x0010000000000000ull; e = be - 1075; }
    u128 p = (u128)m53 * (u128)m53;
    return ext_round128(p, 2 * e);
}

// RN64(a + b), a,b >= 0
PW_HD inline Ext64 ext_add(Ext64 a, Ext64 b) {
    if (a.m == 0) return b;
    if (b.m == 0) return a;
    if (a.e < b.e) { Ext64 t = a; a = b; b = t; }
    int d = a.e - b.e;
    // put a at bits 126..63 (one bit of headroom), b aligned below it
    u128 va = ((u128)a.m) << 63;
    u128 vb;
    bool sticky = false;
    if (d >= 127) {
        vb = 0;
        sticky = true;
    } else {
        u128 full = ((u128)b.m) << 63;
        vb = full >> d;
        if (d > 0 && (full & ((((u128)1) << d) - 1)) != 0) sticky = true;
    }
    u128 s = va + vb;
    // fold the sticky information into the lowest bit below the rounding point:
    // the rounding position is at least 62 bits above bit 0, so OR-ing bit 0 keeps
    // "greater than half" / "less than half" decisions exact.
    if (sticky) s |= 1;
    return ext_round128(s, a.e - 63);
}

// RN64(sqrt(a))
PW_HD inline Ext64 ext_sqrt(Ext64 a) {
    Ext64 z; z.m = 0; z.e = 0;
    if (a.m == 0) return z;
    // M = m * 2^k with k in {63,64} chosen so that (e - k) is even and M in [2^126, 2^128)
    int k = ((a.e - 64) & 1) ? 63 : 64;
    u128 M = ((u128)a.m) << k;
    int e2 = (a.e - k) / 2;  // sqrt(value) = sqrt(M) * 2^e2
    // floor sqrt of M: double estimate then integer correction
    double md = (double)a.m * (k == 64 ? 18446744073709551616.0 : 9223372036854775808.0);
    double sd = pw_sqrt(md);
    uint64_t s;
    if (sd >= 18446744073709551615.0) s = 0xffffffffffffffffull;
    else s = (uint64_t)sd;
    // refine: s += (M - s^2) / (2s) using signed 128-bit remainder
    // (no 128-bit division: gfx950 has no __udivti3; the quotient is small, so
    // a double quotient followed by the exact fix-up loops below is enough)
    for (int it = 0; it < 3; ++it) {
        u128 s2 = (u128)s * (u128)s;
        double two_s = 2.0 * (double)s;
        if (s2 > M) {
            u128 diff = s2 - M;
            double dd = (double)(uint64_t)(diff >> 64) * 18446744073709551616.0 + (double)(uint64_t)diff;
            uint64_t q = (uint64_t)(dd / two_s) + 1;
            s -= q;
        } else {
            u128 diff = M - s2;
            double dd = (double)(uint64_t)(diff >> 64) * 18446744073709551616.0 + (double)(uint64_t)diff;
            uint64_t q = (uint64_t)(dd / two_s);
            if (q == 0) break;
            if (s > 0xffffffffffffffffull - q) s = 0xffffffffffffffffull;
            else s += q;
        }
    }
    while ((u128)s * (u128)s > M) --s;
    while (s != 0xffffffffffffffffull && (u128)(s + 1) * (u128)(s + 1) <= M) ++s;
    // s = floor(sqrt(M)), 2^63 <= s < 2^64.  Round to nearest: up iff M > s^2 + s.
    u128 r = M - (u128)s * (u128)s;
    Ext64 out;
    out.e = e2;
    if (r > (u128)s) {
        s += 1;
        if (s == 0) { s = 0x8000000000000000ull; out.e += 1; }
    }
    out.m = s;
    return out;
}

// RN53(a) as a double
PW_HD inline double ext_to_double(Ext64 a) {
    if (a.m == 0) return 0.0;
    uint64_t m = a.m >> 11;
    uint64_t rem = a.m & 0x7ff;
    int e = a.e + 11;
    if (rem > 0x400 || (rem == 0x400 && (m & 1))) {
        m += 1;
        if (m == (1ull << 53)) { m >>= 1; e += 1; }
    }
    // m in [2^52, 2^53): value = m * 2^e
    union { double d; uint64_t u; } c;
    uint64_t be = (uint64_t)(e + 1075);
    c.u = (be << 52) | (m & 0x000fffffffffffffull);
    return c.d;
}

// dnrm2 for short unit-stride vectors: one extended accumulator, sequential.  The integer emulation,
// always right and about a thousand instructions on the GPU.
PW_NOINLINE PW_HD inline double b_dnrm2_exact(int n, const double* x) {
    Ext64 s; s.m = 0; s.e = 0;
    for (int i = 0; i < n; ++i) s = ext_add(s, ext_square(x[i]));
    return ext_to_double(ext_sqrt(s));
}

// The same value by a short floating-point route whenever that is provably safe (it is for 99 % of
// the arguments), the emulation otherwise.  The x87 result is RN53(RN64(sqrt(S))) with S the sum of
// squares accumulated with a rounding to 64 bits after every operation: at most n squares, n - 1
// additions and one square root, each off by at most 2^-64 relative, so that value lies within
// (n + 1) * 2^-63 (relative) of y = sqrt(x_1^2 + ... + x_n^2) exactly.  y is computed here to about
// 2^-100 in double-double arithmetic (error-free products and sums through fma); when y is farther
// than 2^-58 * y from every midpoint between neighbouring doubles -- n <= 8 keeps (n + 1) 2^-63
// below 2^-59 -- every number that close to y rounds to the same double, which is therefore the
// x87 result.  Checked against the emulation on random and on adversarial (near-midpoint) arguments
// (tests/test_blas_emulation.py); the L-BFGS-B lockstep tests run through it as well.
PW_HD inline double b_dnrm2(int n, const double* x) {
#ifdef PW_NO_FAST_NRM2
    return b_dnrm2_exact(n, x);
#endif
    if (n > 8) return b_dnrm2_exact(n, x);
    // exact sum of the squares as hi + lo
    double hi = 0.0, lo = 0.0;
    for (int i = 0; i < n; ++i) {
        const double a = x[i];
        const double ph = a * a;
        const double pl = pw_fma(a, a, -ph);          // a * a = ph + pl exactly
        const double s = hi + ph;                      // two-sum of hi and ph
        const double bb = s - hi;
        const double err = (hi - (s - bb)) + (ph - bb);
        lo = (lo + pl) + err;
        hi = s;
    }
    if (!(hi > 0.0) || !(hi < 1e300) || hi < 1e-290) return b_dnrm2_exact(n, x);   // zeros, specials, extremes
    {
        const double s = hi + lo;                      // renormalise (|lo| << hi)
        lo = lo - (s - hi);
        hi = s;
    }
    // y = sqrt(hi + lo) = yh + yl: yh the rounded root of hi, yl from the exact residual
    const double yh = pw_sqrt(hi);
    const double qh = yh * yh;
    const double ql = pw_fma(yh, yh, -qh);
    const double res = ((hi - qh) - ql) + lo;          // (hi - qh is exact: both are within an ulp of each other)
    const double yl = res / (yh + yh);
    // r = RN53(yh + yl) and the exact rounding error e of that sum
    const double r = yh + yl;
    const double e = yl - (r - yh);
    // half an ulp of r, from its exponent; powers of two (different ulps on the two sides) go the slow way
    union { double d; uint64_t u; } c;
    c.d = r;
    if ((c.u & 0x000fffffffffffffull) == 0) return b_dnrm2_exact(n, x);
    c.u = (c.u & 0x7ff0000000000000ull) - (53ull << 52);   // 2^(E - 53)
    const double half_ulp = c.d;
    if (half_ulp - pw_abs(e) > r * 3.469446951953614e-18)  // 2^-58
        return r;
    return b_dnrm2_exact(n, x);
}

}  // namespace pw
