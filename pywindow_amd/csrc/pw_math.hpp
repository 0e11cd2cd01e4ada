// pw_math.hpp -- deterministic elementary functions shared by host tests and
// gfx950.  The reference calls numpy's sin/cos/arccos/log10 (utilities.py:1097,
// 1261-1279, 1409-1423); on the reference machine numpy's sin/cos agree with
// glibc, which is correctly rounded for 99.8 % of the arguments that occur
// (golden-spiral angles k*pi*(3-sqrt 5), k < 2000), so the functions here
// evaluate in double-double and round once -- identical on CPU and GPU, and
// within one ulp (usually zero) of the reference's values.  They are NOT the
// device's libm: ocml's results differ from glibc's in the last bit too often.
#pragma once
#include "pw_common.hpp"

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

// 1/n! for n = 2..27 (double-double)
PW_HD inline DD inv_fact(int n) {
    const double T[26][2] = {
        {0.5, 0.0},
        {0.16666666666666666, 9.25185853854297e-18},
        {0.041666666666666664, 2.3129646346357427e-18},
        {0.008333333333333333, 1.1564823173178714e-19},
        {0.001388888888888889, -5.300543954373577e-20},
        {0.0001984126984126984, 1.7209558293420705e-22},
        {2.48015873015873e-05, 2.1511947866775882e-23},
        {2.7557319223985893e-06, -1.858393274046472e-22},
        {2.755731922398589e-07, 2.3767714622250297e-23},
        {2.505210838544172e-08, -1.448814070935912e-24},
        {2.08767569878681e-09, -1.20734505911326e-25},
        {1.6059043836821613e-10, 1.2585294588752098e-26},
        {1.1470745597729725e-11, 2.0655512752830745e-28},
        {7.647163731819816e-13, 7.03872877733453e-30},
        {4.779477332387385e-14, 4.399205485834081e-31},
        {2.8114572543455206e-15, 1.6508842730861433e-31},
        {1.5619206968586225e-16, 1.1910679660273754e-32},
        {8.22063524662433e-18, 2.2141894119604265e-34},
        {4.110317623312165e-19, 1.4412973378659527e-36},
        {1.9572941063391263e-20, -1.3643503830087908e-36},
        {8.896791392450574e-22, -7.911402614872376e-38},
        {3.868170170630684e-23, -8.843177655482344e-40},
        {1.6117375710961184e-24, -3.6846573564509766e-41},
        {6.446950284384474e-26, -1.9330404233703465e-42},
        {2.4795962632247976e-27, -1.2953730964765229e-43},
        {9.183689863795546e-29, 1.4303150396787322e-45},
    };
    return DD{T[n - 2][0], T[n - 2][1]};
}

// sin and cos of a double argument as double-doubles.  |x| < ~1e5.
PW_NOINLINE PW_HD inline void sincos_dd(double x, DD* s_out, DD* c_out) {
    // pi/2 in 33-bit pieces: k * piece is exact for |k| < 2^20
    const double P1 = 1.5707963267341256, P2 = 6.077100506303966e-11,
                 P3 = 2.0222662487111665e-21, P4 = 8.478427660348229e-32,
                 P5 = 2.0670321098263988e-43;
    const double INV_PIO2 = 0.6366197723675814;
    double kd = __builtin_floor(x * INV_PIO2 + 0.5);
    long k = (long)kd;
    DD r = DD{x - kd * P1, 0.0};            // exact (Sterbenz)
    r = dd_add_d(r, -(kd * P2));            // exact products
    r = dd_add_d(r, -(kd * P3));
    r = dd_add(r, dd_neg(two_prod(kd, P4)));
    r = dd_add_d(r, -(kd * P5));
    DD r2 = dd_mul(r, r);
    // sin r = r * (1 - r^2/3! + r^4/5! - ...), cos r = 1 - r^2/2! + r^4/4! - ...
    DD ps = inv_fact(27);
    DD pc = inv_fact(26);
#pragma unroll
    for (int n = 25; n >= 3; n -= 2) {   // fully unrolled: the coefficients become immediates
        ps = dd_add(inv_fact(n), dd_neg(dd_mul(ps, r2)));
        pc = dd_add(inv_fact(n - 1), dd_neg(dd_mul(pc, r2)));
    }
    // ps = 1/3! - r^2/5! + ...,  pc = 1/2! - r^2/4! + ...
    DD sr = dd_add(r, dd_neg(dd_mul(dd_mul(ps, r2), r)));
    DD cr = dd_add(DD{1.0, 0.0}, dd_neg(dd_mul(pc, r2)));
    int q = (int)(k & 3);
    DD s, c;
    if (q == 0) { s = sr; c = cr; }
    else if (q == 1) { s = cr; c = dd_neg(sr); }
    else if (q == 2) { s = dd_neg(sr); c = dd_neg(cr); }
    else { s = dd_neg(cr); c = sr; }
    *s_out = s;
    *c_out = c;
}

PW_HD inline void pw_sincos(double x, double* s, double* c) {
    DD sd, cd;
    sincos_dd(x, &sd, &cd);
    *s = sd.hi + sd.lo;
    *c = cd.hi + cd.lo;
}
PW_HD inline double pw_sin(double x) { double s, c; pw_sincos(x, &s, &c); return s; }
PW_HD inline double pw_cos(double x) { double s, c; pw_sincos(x, &s, &c); return c; }

// arccos on [0, 1] (the reference only ever takes arccos of an absolute cosine,
// utilities.py:1093-1097): Newton on cos(y) = x, last step in double-double.
PW_HD inline double pw_acos01(double x) {
    if (x >= 1.0) return 0.0;
    if (x <= 0.0) return 1.5707963267948966;
    // Abramowitz & Stegun 4.4.45 start (|error| < 7e-5)
    double y = pw_sqrt(1.0 - x) *
               (1.5707288 + x * (-0.2121144 + x * (0.0742610 + x * (-0.0187293))));
    // every step evaluates cos(y) - x in double-double: near x = 1 the difference
    // is O(y^2) and would be lost in a plain double subtraction
    for (int it = 0; it < 3; ++it) {
        DD sd, cd;
        sincos_dd(y, &sd, &cd);
        DD num = dd_add_d(cd, -x);
        double corr = (num.hi + num.lo) / (sd.hi + sd.lo);
        y = y + corr;
    }
    return y;
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
