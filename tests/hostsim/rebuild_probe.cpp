// tests/hostsim/rebuild_probe.cpp -- runs pywindow_amd/csrc/pw_rebuild.hpp on the HOST with a
// one-thread team (see unit_probe.cpp).  Test infrastructure only.
#include "../../pywindow_amd/csrc/pw_rebuild.hpp"
#include <stdlib.h>
#include <string.h>
using namespace pw;
extern "C" int hs_discrete_molecules(int n, const double* xyz, const double* lattice, const double* lattice_inv,
                                     const double* cov, const double* mass, const unsigned char* terminal,
                                     double max_dist, double tol, int rebuild, int atoms_cap, int mols_cap,
                                     int* n_mol, int* status, int* mol_offset, int* src_atom,
                                     signed char* src_image, double* out_xyz, int with_bits) {
    size_t bytes = RebuildWs::bytes(n, rebuild, 1);
    unsigned char* base = (unsigned char*)aligned_alloc(64, (bytes + 63) & ~(size_t)63);
    if (!base) return -5;
    memset(base, 0, bytes);
    RebuildWs* w = RebuildWs::carve(base, n, rebuild, 1);
    // with_bits: bit 0 = the visit bit sets, bit 1 = the scan coordinates in "team-shared" memory
    size_t fb = RebuildWs::fast_bytes(n, rebuild, (with_bits & 1) != 0, (with_bits & 2) != 0);
    unsigned char* fast = (unsigned char*)aligned_alloc(64, (fb + 63) & ~(size_t)63);
    if (!fast) { free(base); return -5; }
    memset(fast, 0xff, fb);                     // the kernel must not rely on zeroed team-shared memory
    w->attach_fast(fast, n, rebuild, (with_bits & 1) != 0, (with_bits & 2) != 0);
    RebuildFrame fr;
    fr.n = n; fr.periodic = lattice != nullptr; fr.rebuild = rebuild;
    fr.xyz = xyz; fr.lattice = lattice; fr.lattice_inv = lattice_inv;
    fr.cov = cov; fr.mass = mass; fr.terminal = terminal; fr.max_dist = max_dist; fr.tol = tol;
    RebuildOut out;
    out.n_mol = n_mol; out.status = status; out.mol_offset = mol_offset; out.src_atom = src_atom;
    out.src_image = src_image; out.xyz = out_xyz; out.atoms_cap = atoms_cap; out.mols_cap = mols_cap;
    rebuild_frame<HostTeam>(fr, *w, out);
    free(fast);
    free(base);
    return 0;
}
