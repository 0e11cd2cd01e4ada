// tests/hostsim/math_probe.cpp -- exposes pw_math.hpp to Python tests.
#include "../../pywindow_amd/csrc/pw_math.hpp"
extern "C" {
void hs_sincos(int n, const double* x, double* s, double* c) { for (int i = 0; i < n; ++i) pw::pw_sincos(x[i], s + i, c + i); }
void hs_acos01(int n, const double* x, double* y) { for (int i = 0; i < n; ++i) y[i] = pw::pw_acos01(x[i]); }
void hs_log10(int n, const double* x, double* y) { for (int i = 0; i < n; ++i) y[i] = pw::pw_log10(x[i]); }
}
