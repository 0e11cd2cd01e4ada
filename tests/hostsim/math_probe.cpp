// tests/hostsim/math_probe.cpp -- exposes pw_math.hpp to Python tests.
#include "../../pywindow_amd/csrc/pw_math.hpp"
extern "C" {
void hs_sincos(int n, const double* x, double* s, double* c) { for (int i = 0; i < n; ++i) pw::pw_sincos(x[i], s + i, c + i); }
void hs_acos(int n, const double* x, double* y) {
    static unsigned tab[65536];
    static bool ready = false;
    if (!ready) { pw::rsqrt14_decode(tab); ready = true; }
    for (int i = 0; i < n; ++i) y[i] = pw::pw_acos_np(x[i], tab);
}
void hs_pow(int n, const double* x, double y, double* out) { for (int i = 0; i < n; ++i) out[i] = pw::pw_pow_np(x[i], y); }
void hs_log10(int n, const double* x, double* y) { for (int i = 0; i < n; ++i) y[i] = pw::pw_log10(x[i]); }
}
