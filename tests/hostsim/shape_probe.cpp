// tests/hostsim/shape_probe.cpp -- runs pywindow_amd/csrc/pw_shape.hpp on the HOST with a
// one-thread team so the summation orders can be checked against the golden vectors without
// a GPU.  Test infrastructure only: the product library never links or calls this.
#include "../../pywindow_amd/csrc/pw_shape.hpp"
using namespace pw;
extern "C" int hs_shape_batch(long n_units, const long* off, const double* xyz, const double* mass,
                              pw_shape_out* out) {
    static ShapeScratch sc;
    for (long u = 0; u < n_units; ++u)
        shape_unit<HostTeam>(sc, xyz + 3 * off[u], mass + off[u], (int)(off[u + 1] - off[u]), out + u);
    return 0;
}
extern "C" int hs_circumcircle(const double* xyz, const int* sets, long n_sets, double* diameter, double* centre) {
    for (long k = 0; k < n_sets; ++k) circumcircle_one(xyz, sets + 3 * k, diameter + k, centre + 3 * k);
    return 0;
}
