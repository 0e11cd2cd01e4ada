// pw_math.hpp -- the elementary functions of the path, restated so that host tests and
// gfx950 produce the SAME BITS as the reference's numpy calls (utilities.py:1097, 1261-1279,
// 1409-1423): numpy.sin / numpy.cos are the C library's (glibc 2.35, s_sin.c), numpy.arccos is
// Intel SVML's __svml_acos8_ha, numpy.log10 only feeds an int() and is reproduced through its
// floor.  None of them is correctly rounded, and the window optimisers amplify a last-bit
// difference in the rotation matrix to 1e-9..1e-6 in a window diameter, so "within one ulp" is
// not enough: each routine below follows the original operation by operation, fused
// multiply-adds included.  They are NOT the device's libm (ocml differs in the last bit).
#pragma once
#include "pw_common.hpp"
#include "pw_rsqrt14_data.hpp"
#include "pw_sincos_data.hpp"
#include "pw_pow_data.hpp"

namespace pw {

struct DD {
    double hi, lo;
};

PW_HD inline DD two_sum(double a, double b) {
    double s = a + b;
    double bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return DD{s, e};
}
PW_HD inline DD quick_two_sum(double a, double b) {  // |a| >= |b|
    double s = a + b;
    double e = b - (s - a);
    return DD{s, e};
}
PW_HD inline DD two_prod(double a, double b) {
    double p = a * b;
    double e = pw_fma(a, b, -p);
    return DD{p, e};
}
PW_HD inline DD dd_add(DD a, DD b) {
    DD s = two_sum(a.hi, b.hi);
    DD t = two_sum(a.lo, b.lo);
    s.lo = s.lo + t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo = s.lo + t.lo;
    return quick_two_sum(s.hi, s.lo);
}
PW_HD inline DD dd_add_d(DD a, double b) {
    DD s = two_sum(a.hi, b);
    s.lo = s.lo + a.lo;
    return quick_two_sum(s.hi, s.lo);
}
PW_HD inline DD dd_mul(DD a, DD b) {
    DD p = two_prod(a.hi, b.hi);
    p.lo = p.lo + (a.hi * b.lo + a.lo * b.hi);
    return quick_two_sum(p.hi, p.lo);
}
PW_HD inline DD dd_mul_d(DD a, double b) {
    DD p = two_prod(a.hi, b);
    p.lo = p.lo + a.lo * b;
    return quick_two_sum(p.hi, p.lo);
}
PW_HD inline DD dd_neg(DD a) { return DD{-a.hi, -a.lo}; }

PW_HD inline double pw_bits2d(uint64_t u) { union { double d; uint64_t u; } cv; cv.u = u; return cv.d; }
PW_HD inline uint64_t pw_d2bits(double d) { union { double d; uint64_t u; } cv; cv.d = d; return cv.u; }

// ---- numpy.sin / numpy.cos for float64 ------------------------------------------------------------
// numpy calls the C library for these; on the reference's platform that is glibc 2.35's
// s_sin.c (IBM Accurate Mathematical Library branch; < 0.55 ulp, i.e. NOT always correctly
// rounded: 0.2 % of the golden-spiral angles differ from the correctly rounded value).  The
// routine is restated with the fused multiply-adds exactly where the FMA build of glibc has
// them (x86-64 ifunc variant used on every AVX2 machine): table look-up of sin/cos(k/128),
// short Taylor polynomials for the remainder, Cody-Waite reduction in four parts for
// 2.43 < |x| < 1.05e8.  Checked against the C library on millions of arguments
// (tests/test_math.py).  Arguments beyond 1.05e8 never occur on the path.
PW_HD inline double sc_copysign(double mag, double sgn) {
    return pw_bits2d((pw_d2bits(mag) & 0x7fffffffffffffffull) | (pw_d2bits(sgn) & 0x8000000000000000ull));
}
// TAYLOR_SIN(xx, x, dx), s_sin.c
PW_HD inline double sc_taylor_sin(double x, double dx) {
    double xx = x * x;
    double p = pw_fma(xx, SC_S5, SC_S4);
    p = pw_fma(xx, p, SC_S3);
    p = pw_fma(xx, p, SC_S2);
    p = pw_fma(xx, p, SC_S1);
    double t = pw_fma(xx, pw_fma(p, x, -(0.5 * dx)), dx);
    return x + t;
}
// do_sin(x, dx), s_sin.c: sin(x + dx), |x| < 0.86
PW_HD inline double sc_do_sin(double x, double dx) {
    const double xold = x;
    if (pw_abs(x) < SC_TAYLOR_MAX) return sc_taylor_sin(x, dx);
    if (x <= 0.0) dx = -dx;
    const double u = SC_BIG + pw_abs(x);
    x = pw_abs(x) - (u - SC_BIG);
    const int k = (int)(unsigned)(pw_d2bits(u) & 0xffffffffull) * 4;
    const double xx = x * x;
    const double s = x + pw_fma(x * xx, pw_fma(xx, SC_SN5, SC_SN3), dx);
    const double c = pw_fma(x, dx, xx * pw_fma(xx, pw_fma(xx, SC_CS6, SC_CS4), SC_CS2));
    const double sn = SC_TAB[k], ssn = SC_TAB[k + 1], cs = SC_TAB[k + 2], ccs = SC_TAB[k + 3];
    const double cor = pw_fma(s, cs, pw_fma(-c, sn, pw_fma(s, ccs, ssn)));
    return sc_copysign(sn + cor, xold);
}
// do_cos(x, dx), s_sin.c: cos(x + dx), |x| < 0.86
PW_HD inline double sc_do_cos(double x, double dx) {
    if (x < 0.0) dx = -dx;
    const double u = SC_BIG + pw_abs(x);
    x = pw_abs(x) - (u - SC_BIG) + dx;
    const int k = (int)(unsigned)(pw_d2bits(u) & 0xffffffffull) * 4;
    const double xx = x * x;
    const double s = pw_fma(x * xx, pw_fma(xx, SC_SN5, SC_SN3), x);
    const double c = xx * pw_fma(xx, pw_fma(xx, SC_CS6, SC_CS4), SC_CS2);
    const double sn = SC_TAB[k], ssn = SC_TAB[k + 1], cs = SC_TAB[k + 2], ccs = SC_TAB[k + 3];
    const double cor = pw_fma(-s, sn, pw_fma(-c, cs, pw_fma(-s, ssn, ccs)));
    return cs + cor;
}
// reduce_sincos(x, &a, &da), s_sin.c: x = n*pi/2 + a + da; returns n mod 4
PW_HD inline int sc_reduce(double x, double* a, double* da) {
    const double t = pw_fma(x, SC_HPINV, SC_TOINT);
    const double xn = t - SC_TOINT;
    const double y = pw_fma(-xn, SC_MP2, pw_fma(-xn, SC_MP1, x));
    const int n = (int)(pw_d2bits(t) & 3ull);
    const double t2 = pw_fma(-xn, SC_PP3, y);
    double db = pw_fma(-xn, SC_PP3, y - t2);
    const double b = pw_fma(-xn, SC_PP4, t2);
    db = db + pw_fma(-xn, SC_PP4, t2 - b);
    *a = b;
    *da = db;
    return n;
}
PW_HD inline double sc_do_sincos(double a, double da, int n) {
    double r = (n & 1) ? sc_do_cos(a, da) : sc_do_sin(a, da);
    return (n & 2) ? -r : r;
}
// libm sin(x) as numpy.sin sees it
PW_NOINLINE PW_HD inline double pw_sin_np(double x) {
    const unsigned k = (unsigned)(pw_d2bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e500000u) return x;
    if (k < 0x3feb6000u) return sc_do_sin(x, 0.0);
    if (k < 0x400368fdu) {
        double t = SC_HP0 - pw_abs(x);
        return sc_copysign(sc_do_cos(t, SC_HP1), x);
    }
    double a, da;
    int n = sc_reduce(x, &a, &da);
    return sc_do_sincos(a, da, n);
}
// libm cos(x) as numpy.cos sees it
PW_NOINLINE PW_HD inline double pw_cos_np(double x) {
    const unsigned k = (unsigned)(pw_d2bits(x) >> 32) & 0x7fffffffu;
    if (k < 0x3e400000u) return 1.0;
    if (k < 0x3feb6000u) return sc_do_cos(x, 0.0);
    if (k < 0x400368fdu) {
        double y = SC_HP0 - pw_abs(x);
        double a = y + SC_HP1;
        double da = (y - a) + SC_HP1;
        return sc_do_sin(a, da);
    }
    double a, da;
    int n = sc_reduce(x, &a, &da);
    return sc_do_sincos(a, da, n + 1);
}

PW_HD inline void pw_sincos(double x, double* s, double* c) {
    *s = pw_sin_np(x);
    *c = pw_cos_np(x);
}
PW_HD inline double pw_sin(double x) { return pw_sin_np(x); }
PW_HD inline double pw_cos(double x) { return pw_cos_np(x); }

// ---- float64 scalar ** (numpy -> the C library's pow) --------------------------------------------
// The reference writes x ** 2, r ** 3 and m ** 0.5 on numpy float64 SCALARS (utilities.py:93, 431,
// 1095, 1434); those go to glibc's pow() (e_pow.c: table-driven log to ~68 bits, then exp), which
// is within 0.52 ulp but not correctly rounded: pow(x, 2.0) != x*x for 0.08 % of the arguments.
// Main path of the FMA build, operation by operation; x positive and normal, |y log x| moderate
// (everything on this path), otherwise the caller's plain expression is used.
PW_NOINLINE PW_HD inline double pw_pow_np(double x, double y) {
    const uint64_t ix = pw_d2bits(x);
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 0x7f);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = pw_bits2d(ix - (tmp & 0xfff0000000000000ull));
    const double kd = (double)k;
    const double ln2hi = POW_LOG_HEAD[0], ln2lo = POW_LOG_HEAD[1];
    const double A0 = POW_LOG_HEAD[2], A1 = POW_LOG_HEAD[3], A2 = POW_LOG_HEAD[4], A3 = POW_LOG_HEAD[5],
                 A4 = POW_LOG_HEAD[6], A5 = POW_LOG_HEAD[7], A6 = POW_LOG_HEAD[8];
    const double invc = POW_LOG_TAB[4 * i], logc = POW_LOG_TAB[4 * i + 2], logctail = POW_LOG_TAB[4 * i + 3];
    // log(x) = k ln2 + log(c) + log1p(z/c - 1), as hi + lo
    const double t1 = pw_fma(kd, ln2hi, logc);
    const double r = pw_fma(z, invc, -1.0);
    const double ar = r * A0;
    const double lo1 = pw_fma(kd, ln2lo, logctail);
    const double p12 = pw_fma(r, A2, A1);
    const double p34 = pw_fma(r, A4, A3);
    const double t2 = r + t1;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double lo3 = pw_fma(ar, r, -ar2);
    const double lo2 = (t1 - t2) + r;
    const double p56 = pw_fma(r, A6, A5);
    const double hi = t2 + ar2;
    const double lo4 = (t2 - hi) + ar2;
    const double inner = pw_fma(ar2, pw_fma(p56, ar2, p34), p12);
    double lo = lo1 + lo2;
    lo = lo + lo3;
    lo = lo + lo4;
    lo = pw_fma(ar3, inner, lo);
    const double lhi = hi + lo;
    const double llo = (hi - lhi) + lo;
    // exp(y * log x)
    const double ehi = y * lhi;
    const double elo = pw_fma(y, llo, pw_fma(lhi, y, -ehi));
    const double InvLn2N = POW_EXP_HEAD[0], Shift = POW_EXP_HEAD[1], NegLn2hiN = POW_EXP_HEAD[2],
                 NegLn2loN = POW_EXP_HEAD[3], C2 = POW_EXP_HEAD[4], C3 = POW_EXP_HEAD[5], C4 = POW_EXP_HEAD[6],
                 C5 = POW_EXP_HEAD[7];
    const double zz = pw_fma(ehi, InvLn2N, Shift);
    const uint64_t ki = pw_d2bits(zz);
    const double kdd = zz - Shift;
    double rr = pw_fma(kdd, NegLn2loN, pw_fma(kdd, NegLn2hiN, ehi));
    const int idx = 2 * (int)(ki & 0x7f);
    const uint64_t sbits = POW_EXP_TAB[idx + 1] + (ki << 45);
    rr = elo + rr;
    const double q23 = pw_fma(rr, C3, C2);
    const double tail_r = rr + pw_bits2d(POW_EXP_TAB[idx]);
    const double r2 = rr * rr;
    const double q45 = pw_fma(rr, C5, C4);
    const double acc = pw_fma(q23, r2, tail_r);
    const double r4 = r2 * r2;
    const double tmpv = pw_fma(q45, r4, acc);
    const double scale = pw_bits2d(sbits);
    return pw_fma(tmpv, scale, scale);
}

// x ** 2 and x ** 3 for any finite x (pow's sign handling for integer exponents)
PW_HD inline double pw_square_np(double x) {
    double a = pw_abs(x);
    if (!(a >= 2.2250738585072014e-308 && a < 1e150)) return x * x;
    return pw_pow_np(a, 2.0);
}
PW_HD inline double pw_cube_np(double x) {
    double a = pw_abs(x);
    if (!(a >= 2.2250738585072014e-308 && a < 1e100)) return x * x * x;
    double r = pw_pow_np(a, 3.0);
    return x < 0.0 ? -r : r;
}

// ---- numpy.arccos for float64 ------------------------------------------------------------------
// numpy evaluates arccos with Intel SVML (__svml_acos8_ha; also for scalars: the AVX-512 loop
// handles the tail with masks).  Its results differ from the correctly rounded arccos in 9 % of
// the arguments, so the routine is restated operation by operation (FMA placement included)
// from the vector code: for x^2 < y = (1-|x|)/2 a degree-12 polynomial in x^2, otherwise
// 2*asin(sqrt(y)) with sqrt(2y) as a hi/lo pair refined from the VRSQRT14PD estimate.  That
// instruction is a function of the exponent parity and the top 15 mantissa bits only; its 65536
// values (identical on Intel Xeon and AMD Zen 5) are in pw_rsqrt14_data.hpp.  Checked against
// numpy on 4e6 arguments without a mismatch (tests/test_math.py).
//
// `tab`: 65536 entries decoded by rsqrt14_decode(): (result bits >> 36) for operands in
// [0.5, 1) then [1, 2), indexed by the top 15 mantissa bits.
inline void rsqrt14_decode(unsigned* tab) {
    const unsigned start[2] = {RSQRT14_START_0, RSQRT14_START_1};
    const char* delta[2] = {RSQRT14_DELTA_0, RSQRT14_DELTA_1};
    for (int p = 0; p < 2; ++p) {
        unsigned v = start[p];
        for (int i = 0; i < 32768; ++i) {
            tab[p * 32768 + i] = v;
            if (i < 32767) {
                char c = delta[p][i];
                v -= (unsigned)(c <= '9' ? c - '0' : c - 'a' + 10);
            }
        }
    }
}
// VRSQRT14PD for a positive normal operand
PW_HD inline double pw_rsqrt14(double y, const unsigned* tab) {
    uint64_t u = pw_d2bits(y);
    int ef = (int)((u >> 52) & 0x7ff);
    int p = ef & 1;
    int k = (ef - (1022 + p)) / 2;                 // y = y' * 4^k with y' in [0.5, 2)
    unsigned v = tab[p * 32768 + (int)((u >> 37) & 0x7fff)];
    uint64_t rb = ((uint64_t)((int)(v >> 16) - k) << 52) | ((uint64_t)(v & 0xffffu) << 36);
    return pw_bits2d(rb);
}
// numpy.arccos(x) for |x| <= 1
PW_NOINLINE PW_HD inline double pw_acos_np(double x, const unsigned* tab) {
    const double c4 = pw_bits2d(0xbf918000993b24c3ull), c3 = pw_bits2d(0x3fa400006f70d42dull),
                 c2 = pw_bits2d(0xbfb7fffffffffe97ull), c1 = pw_bits2d(0x3fcfffffffffff9dull);
    const double q12 = pw_bits2d(0x3fa07520c70eb909ull), q11 = pw_bits2d(0xbf90fb17f7dbb0edull),
                 q10 = pw_bits2d(0x3f943f44bfbc3baeull), q9 = pw_bits2d(0x3f7a583395d45ed5ull),
                 q8 = pw_bits2d(0x3f88f8dc2afccad6ull), q7 = pw_bits2d(0x3f8c6dbbcb88bd57ull),
                 q6 = pw_bits2d(0x3f91c6dcf538ad2eull), q5 = pw_bits2d(0x3f96e89cebdefaddull),
                 q4 = pw_bits2d(0x3f9f1c72e13ad8beull), q3 = pw_bits2d(0x3fa6db6db3b445f8ull),
                 q2 = pw_bits2d(0x3fb333333337e0deull), q1 = pw_bits2d(0x3fc555555555529cull);
    const double pi_lo = pw_bits2d(0x3ca1a62633145c07ull), pi_hi = pw_bits2d(0x400921fb54442d18ull),
                 pio2_lo = pw_bits2d(0x3c91a62633145c07ull), pio2_hi = pw_bits2d(0x3ff921fb54442d18ull);
    const uint64_t sign = pw_d2bits(x) & 0x8000000000000000ull;
    const double nax = pw_bits2d(pw_d2bits(x) | 0x8000000000000000ull);   // -|x|
    const double y = pw_fma(0.5, nax, 0.5);
    const double x2 = nax * nax;
    const bool tiny = y < pw_bits2d(0x3000000000000000ull);
    const double r = tiny ? 0.0 : pw_rsqrt14(y, tab);
    const double t = (x2 < y) ? x2 : y;
    const double y2 = y + y;
    const bool root = !(t < y);                    // 2*asin(sqrt(y)) branch
    const bool reflect = root && !(t < x);         // ... of a negative argument: pi - (...)
    const double rr = r * r;
    const double S = y2 * r;                       // ~ sqrt(2y) * sqrt(2): 2*sqrt(y), high part
    const double e = pw_fma(rr, y2, -2.0);
    const double Slo = pw_fma(r, y2, -S);
    double g = pw_fma(c4, e, c3);
    const double Se = S * e;
    g = pw_fma(e, g, c2);
    g = pw_fma(e, g, c1);
    const double a0 = pw_fma(q10, t, q9);
    double C = pw_fma(Se, g, -Slo);                // 2*sqrt(y) = S - C
    double a1 = pw_fma(q12, t, q11);
    const double a3 = pw_fma(q4, t, q3);
    const double t2 = t * t;
    double a2 = pw_fma(q8, t, q7);
    a1 = pw_fma(t2, a1, a0);
    const double t4 = t2 * t2;
    const double a4 = pw_fma(q6, t, q5);
    a2 = pw_fma(t2, a2, a4);
    a1 = pw_fma(t4, a1, a2);
    a1 = pw_fma(t2, a1, a3);
    if (!root) C = 0.0;
    a1 = pw_fma(t, a1, q2);
    a1 = pw_fma(t, a1, q1);
    const double Q = a1 * t;
    double lo = root ? 0.0 : pio2_lo, hi = root ? 0.0 : pio2_hi;
    if (reflect) { lo = pi_lo; hi = pi_hi; }
    const double slo = pw_bits2d(pw_d2bits(lo) ^ sign);
    const double H = root ? S : nax;
    const double z6 = slo - C;
    const double z3 = H - C;
    double z = pw_fma(z3, Q, z6);
    z = z + H;
    return pw_bits2d(pw_d2bits(z) ^ sign) + hi;
}

// natural log, fdlibm-style kernel (< 1 ulp); x > 0, normal.
PW_HD inline double pw_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    union { double d; uint64_t u; } cv;
    cv.d = x;
    int e = (int)((cv.u >> 52) & 0x7ff) - 1023;
    cv.u = (cv.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;  // m in [1,2)
    double m = cv.d;
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

PW_HD inline double pw_log10(double x) {
    const double ivln10_hi = 0.4342944819032518, ivln10_lo = 1.098319650216765e-17;
    double lg = pw_log(x);
    DD p = two_prod(lg, ivln10_hi);
    return p.hi + (p.lo + lg * ivln10_lo);
}

}  // namespace pw
