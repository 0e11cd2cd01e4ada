// pw_rebuild.hpp -- periodic pre-processing: split one frame of a (periodic) molecular
// system into discrete molecules, optionally re-assembling molecules that were wrapped
// across the cell faces from the 3x3x3 supercell.
//
// Counterpart of the reference's create_supercell (utilities.py:768-810) and
// discrete_molecules (utilities.py:820-1085) as driven by MolecularSystem.rebuild_system /
// make_modular (molecular.py:672-708, 798-824) -- SURVEY.md section 8, row f-1.  One team
// (workgroup) handles one frame.
//
// The reference keeps atoms as Python lists [element, atom_id, x, y, z] with coordinates
// rounded to 8 decimals and compares / removes them BY VALUE; the traversal is a
// breadth-first walk over bonded neighbours whose visiting order fixes the atom order of
// every output molecule (and, through it, every order-sensitive sum downstream).  That
// order is reproduced exactly:
//   * value coordinates: round(x, 8) for the cell atoms, round(M (M^-1 x + shift), 8) for the
//     27 images, with numpy's 3x3 matrix-vector association (rb_mat3) and Python's
//     correctly-rounded decimal rounding (rb_round8);
//   * an atom of the central image whose value equals the cell atom's is the SAME list item
//     (canonical id = the cell atom); all other image atoms are items of their own;
//   * per visited heavy atom the reference scans the remaining cell atoms in index order,
//     then the supercell in (image, atom) order; here only a conservative candidate list is
//     scanned (translation-invariant, built once per frame) and the hits are ordered by
//     that same position key before they are merged;
//   * bond test = the reference's two formulas: scikit-learn's euclidean_distances for the
//     0.1 < d < max_dist pre-filter, distance() (utilities.py:80-93) against Rcov_i + Rcov_j
//     +- tol.
// Not reproduced: value-equality between an image atom and a DIFFERENT cell atom (an input
// that lists the same atom on two opposite cell faces); cells thinner than max_dist.
#pragma once
#include "pw_unit.hpp"

namespace pw {

constexpr int RB_NB_CAP = 32;      // conservative neighbour candidates kept per heavy atom
constexpr int RB_SEG_CAP = 32;     // bonded neighbours one atom can contribute per layer
constexpr int RB_CHUNK = 256;      // atoms of a layer expanded between two merges
constexpr int RB_CENTRAL = 13;     // image (0,0,0) in the a,b,c-nested 3x3x3 enumeration

// status bits of one frame (pw_cell_out.status)
constexpr int RB_ST_NB_OVERFLOW = 1;      // > RB_NB_CAP candidates around one atom
constexpr int RB_ST_SEG_OVERFLOW = 2;     // > RB_SEG_CAP bonded neighbours of one atom
constexpr int RB_ST_ATOMS_OVERFLOW = 4;   // output atom capacity exceeded
constexpr int RB_ST_MOLS_OVERFLOW = 8;    // output molecule capacity exceeded
constexpr int RB_ST_THIN_CELL = 16;       // a cell height is below max_dist: bonds could span two images

PW_HD inline int rb_atomic_add(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    int old = *p;
    *p = old + v;
    return old;
#endif
}
PW_HD inline void rb_atomic_or(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}

// numpy: np.matrix(m) * v.reshape(-1, 1) for a 3x3 m (utilities.py:722-743), as OpenBLAS
// evaluates it: y_i = fma(m_i2, v2, fma(m_i0, v0, m_i1 * v1)).
PW_HD inline void rb_mat3(const double* m, double x, double y, double z, double* out) {
    for (int i = 0; i < 3; ++i)
        out[i] = pw_fma(m[3 * i + 2], z, pw_fma(m[3 * i], x, m[3 * i + 1] * y));
}

// Python round(x, 8) (compose_atom_list, utilities.py:187-220): correctly rounded decimal
// rounding, half-even, and the correctly rounded way back.  x * 1e8 is inexact; its exact
// residual (one fma) settles the ties.
PW_HD inline double rb_round8(double x) {
    double p = x * 1e8;
    if (!(pw_abs(p) < 4503599627370496.0)) return x;
    double e = pw_fma(x, 1e8, -p);
    double f = __builtin_floor(p);
    double r = p - f;
    double k;
    if (r < 0.5) k = f;
    else if (r > 0.5) k = f + 1.0;
    else if (e > 0.0) k = f + 1.0;
    else if (e < 0.0) k = f;
    else k = (f * 0.5 == __builtin_floor(f * 0.5)) ? f : f + 1.0;
    return k / 1e8;
}

// scikit-learn euclidean_distances, N x 1 call shape: row x (with |x|^2 = xx) against point p
PW_HD inline double rb_dist_sk(const double* x, double xx, double px, double py, double pz, double pp) {
    double g = pw_fma(x[2], pz, pw_fma(x[0], px, x[1] * py));
    double d2 = ((-2.0 * g) + xx) + pp;
    return pw_sqrt(d2 > 0.0 ? d2 : 0.0);
}

struct RebuildWs {
    double* V;            // n x 3   value coordinates of the cell atoms
    double* Vxx;          // n
    double* S;            // 27n x 3 value coordinates of the supercell atoms (rebuild only)
    double* msum;         // 28n     masses of the molecule being closed (numpy pairwise sum)
    double* red_v;        // team size: reduction slots
    int* red_i;           // team size
    int* nb_cnt;          // n
    int* nb;              // n x RB_NB_CAP: image * n + atom
    int* stamp_final;     // ids: molecule serial
    int* stamp_temp;      // ids: layer serial
    int* work;            // ids
    int* work_next;       // ids
    int* final_;          // ids
    int* seg_cnt;         // RB_CHUNK
    long long* seg;       // RB_CHUNK x RB_SEG_CAP: position key << 32 | canonical id
    unsigned char* remaining;   // n
    unsigned char* alias;       // n
    double box[27 * 6];
    double com[3], origin[3], bound[2];
    int n_work, n_next, n_final, start, status, n_mol, n_out;

    PW_HD static size_t ids(int n, int rebuild) { return rebuild ? (size_t)28 * n : (size_t)n; }
    PW_HD static size_t bytes(int n, int rebuild, int team) {
        size_t id = ids(n, rebuild);
        size_t d = (size_t)3 * n + n + (rebuild ? (size_t)81 * n : 0) + id + team;
        size_t i = (size_t)team + n + (size_t)n * RB_NB_CAP + 5 * id + RB_CHUNK;
        size_t l = (size_t)RB_CHUNK * RB_SEG_CAP;
        return sizeof(RebuildWs) + 64 + d * 8 + l * 8 + i * 4 + 2 * (size_t)n + 64;
    }
    // `base` -> [RebuildWs header][arrays]; returns the header
    PW_HD static RebuildWs* carve(unsigned char* base, int n, int rebuild, int team) {
        RebuildWs* w = (RebuildWs*)base;
        unsigned char* p = base + ((sizeof(RebuildWs) + 63) & ~(size_t)63);
        size_t id = ids(n, rebuild);
        w->V = (double*)p; p += (size_t)3 * n * 8;
        w->Vxx = (double*)p; p += (size_t)n * 8;
        w->S = (double*)p; p += rebuild ? (size_t)81 * n * 8 : 0;
        w->msum = (double*)p; p += id * 8;
        w->red_v = (double*)p; p += (size_t)team * 8;
        w->seg = (long long*)p; p += (size_t)RB_CHUNK * RB_SEG_CAP * 8;
        w->red_i = (int*)p; p += (size_t)team * 4;
        w->nb_cnt = (int*)p; p += (size_t)n * 4;
        w->nb = (int*)p; p += (size_t)n * RB_NB_CAP * 4;
        w->stamp_final = (int*)p; p += id * 4;
        w->stamp_temp = (int*)p; p += id * 4;
        w->work = (int*)p; p += id * 4;
        w->work_next = (int*)p; p += id * 4;
        w->final_ = (int*)p; p += id * 4;
        w->seg_cnt = (int*)p; p += (size_t)RB_CHUNK * 4;
        w->remaining = p; p += n;
        w->alias = p; p += n;
        return w;
    }
};

struct RebuildFrame {         // inputs of one frame
    int n;
    int periodic;             // lattice given (modes 2, 3 of utilities.py:843-851)
    int rebuild;              // mode 3
    const double* xyz;        // n x 3 as loaded (not rounded)
    const double* lattice;    // 9, row-major; may be null when !periodic
    const double* lattice_inv;
    const double* cov;        // n
    const double* mass;       // n
    const unsigned char* terminal;   // n: element in the reference's `exceptions` list
    double max_dist, tol;
};

struct RebuildOut {           // outputs of one frame
    int* n_mol;               // 1
    int* status;              // 1
    int* mol_offset;          // mols_cap + 1
    int* src_atom;            // atoms_cap: index of the cell atom
    signed char* src_image;   // atoms_cap: -1 = the cell atom itself, else image 0..26
    double* xyz;              // atoms_cap x 3: value coordinates
    int atoms_cap, mols_cap;
};

// node (canonical id) -> cell atom, image offsets, value coordinates
PW_HD inline void rb_decode(const RebuildWs& w, int n, int id, int* q, int* ax, int* ay, int* az,
                            const double** pos) {
    if (id < n) {
        *q = id; *ax = *ay = *az = 0; *pos = &w.V[3 * id];
    } else {
        int s = id - n;
        int img = s / n;
        *q = s - img * n;
        *ax = img / 9 - 1; *ay = (img / 3) % 3 - 1; *az = img % 3 - 1;
        *pos = &w.S[3 * (size_t)s];
    }
}

// one wave expands one atom of the current layer: lanes over its candidate list
template <class T>
PW_HD inline void rb_expand(const RebuildFrame& fr, RebuildWs& w, int id, int slot) {
    const int n = fr.n;
    int q0, ax, ay, az;
    const double* P;
    rb_decode(w, n, id, &q0, &ax, &ay, &az, &P);
    if (fr.terminal[q0]) return;
    const double px = P[0], py = P[1], pz = P[2];
    const double pp = sq3(px, py, pz);
    const double ri = fr.cov[q0];
    const int cnt = w.nb_cnt[q0] < RB_NB_CAP ? w.nb_cnt[q0] : RB_NB_CAP;
    for (int e = T::lane(); e < cnt; e += T::WSIZE) {
        int packed = w.nb[(size_t)q0 * RB_NB_CAP + e];
        int dimg = packed / n;
        int q = packed - dimg * n;
        int bx = ax + dimg / 9 - 1, by = ay + (dimg / 3) % 3 - 1, bz = az + dimg % 3 - 1;
        bool central = bx == 0 && by == 0 && bz == 0;
        double rc = ri + fr.cov[q];
        double lo = rc - fr.tol, hi = rc + fr.tol;
        for (int part = 0; part < 2; ++part) {
            const double* X;
            double xx;
            long long key;
            if (part == 0) {
                // remaining cell atoms (utilities.py:996-1013)
                if (!central || !w.remaining[q]) continue;
                X = &w.V[3 * q];
                xx = w.Vxx[q];
                key = ((long long)q << 32) | (unsigned)q;
            } else {
                // supercell atoms that are not (by value) in the remaining atom list (:1014-1036)
                if (!fr.rebuild) continue;
                if (bx < -1 || bx > 1 || by < -1 || by > 1 || bz < -1 || bz > 1) continue;
                bool same_item = central && w.alias[q];
                if (same_item && w.remaining[q]) continue;
                int s = ((bx + 1) * 9 + (by + 1) * 3 + (bz + 1)) * n + q;
                X = &w.S[3 * (size_t)s];
                xx = sq3(X[0], X[1], X[2]);
                key = ((long long)(n + s) << 32) | (unsigned)(same_item ? q : n + s);
            }
            double d = rb_dist_sk(X, xx, px, py, pz, pp);
            if (!(d > 0.1 && d < fr.max_dist)) continue;
            double dx = px - X[0], dy = py - X[1], dz = pz - X[2];
            double r2 = (dx * dx + dy * dy) + dz * dz;
            double r = r2 >= 2.2250738585072014e-308 ? pw_pow_np(r2, 0.5) : pw_sqrt(r2);   // float ** 0.5: libm pow
            if (!(lo < r && r < hi)) continue;
            int k = rb_atomic_add(&w.seg_cnt[slot], 1);
            if (k < RB_SEG_CAP) w.seg[(size_t)slot * RB_SEG_CAP + k] = key;
            else rb_atomic_or(&w.status, RB_ST_SEG_OVERFLOW);
        }
    }
}

template <class T>
PW_HD inline void rebuild_frame(const RebuildFrame& fr, RebuildWs& w, const RebuildOut& out) {
    const int n = fr.n;
    const int n_ids = (int)RebuildWs::ids(n, fr.rebuild);
    const int tid = T::tid();
    if (tid == 0) {
        w.status = 0; w.n_mol = 0; w.n_out = 0;
        out.mol_offset[0] = 0;
    }
    // ---- value coordinates ------------------------------------------------------------
    for (int i = tid; i < n; i += T::SIZE) {
        double x = rb_round8(fr.xyz[3 * i]), y = rb_round8(fr.xyz[3 * i + 1]), z = rb_round8(fr.xyz[3 * i + 2]);
        w.V[3 * i] = x; w.V[3 * i + 1] = y; w.V[3 * i + 2] = z;
        w.Vxx[i] = sq3(x, y, z);
        w.remaining[i] = 1;
        w.alias[i] = 0;
        w.nb_cnt[i] = 0;
    }
    for (int i = tid; i < n_ids; i += T::SIZE) { w.stamp_final[i] = 0; w.stamp_temp[i] = 0; }
    if (fr.rebuild) {
        // create_supercell: frac = M^-1 x; images a, b, c nested; cart = M (frac + shift)
        for (int i = tid; i < n; i += T::SIZE) {
            double fq[3];
            rb_mat3(fr.lattice_inv, fr.xyz[3 * i], fr.xyz[3 * i + 1], fr.xyz[3 * i + 2], fq);
            for (int img = 0; img < 27; ++img) {
                double sa = (double)(img / 9 - 1), sb = (double)((img / 3) % 3 - 1), sc = (double)(img % 3 - 1);
                double c[3];
                rb_mat3(fr.lattice, fq[0] + sa, fq[1] + sb, fq[2] + sc, c);
                size_t s = (size_t)img * n + i;
                w.S[3 * s] = rb_round8(c[0]); w.S[3 * s + 1] = rb_round8(c[1]); w.S[3 * s + 2] = rb_round8(c[2]);
            }
            size_t s0 = (size_t)RB_CENTRAL * n + i;
            w.alias[i] = (w.S[3 * s0] == w.V[3 * i] && w.S[3 * s0 + 1] == w.V[3 * i + 1] &&
                          w.S[3 * s0 + 2] == w.V[3 * i + 2]) ? 1 : 0;
        }
    }
    T::sync();
    // ---- system centre of mass (utilities.py:127-148 on the unrounded input) -----------
    constexpr int SUM_THREAD = T::SIZE > 3 ? 3 : 0;
    for (int col = tid; col < 3; col += T::SIZE) {
        double acc = 0.0;
        for (int r = 0; r < n; ++r) {
            double t = fr.xyz[3 * r + col] * fr.mass[r];
            acc = r == 0 ? t : acc + t;
        }
        w.com[col] = acc;
    }
    if (tid == SUM_THREAD) w.red_v[0] = np_sum_serial(fr.mass, n);
    if (fr.rebuild) {
        // bounding box of every image (candidate culling only)
        for (int img = tid; img < 27; img += T::SIZE) {
            double lo[3] = {PW_INF, PW_INF, PW_INF}, hi[3] = {-PW_INF, -PW_INF, -PW_INF};
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < 3; ++c) {
                    double v = w.S[3 * ((size_t)img * n + i) + c];
                    lo[c] = pw_min(lo[c], v); hi[c] = pw_max(hi[c], v);
                }
            for (int c = 0; c < 3; ++c) { w.box[6 * img + c] = lo[c]; w.box[6 * img + 3 + c] = hi[c]; }
        }
    }
    T::sync();
    if (tid == 0) {
        double total = w.red_v[0];
        for (int c = 0; c < 3; ++c) w.com[c] = w.com[c] / total;
        if (fr.periodic) {
            // origin skewed by 0.01 along x; pseudo origin at fractional (0.26, 0.25, 0.25)
            // (utilities.py:889-899); <-0.5, 0.5> cell when the system COM is at the origin
            rb_mat3(fr.lattice, 0.26, 0.25, 0.25, w.origin);
            bool centred = pw_abs(w.com[0] - 0.01) <= 1.0 + 1e-5 * 0.01 && pw_abs(w.com[1]) <= 1.0 &&
                           pw_abs(w.com[2]) <= 1.0;
            w.bound[0] = centred ? -0.5 : 0.0;
            w.bound[1] = centred ? 0.5 : 1.0;
            // candidate lists assume a bond never spans two images
            double h[3];
            for (int c = 0; c < 3; ++c) h[c] = pw_abs(fr.lattice[3 * c + c]);
            if (fr.rebuild && (h[0] < fr.max_dist || h[1] < fr.max_dist || h[2] < fr.max_dist))
                w.status |= RB_ST_THIN_CELL;
        } else {
            w.origin[0] = w.com[0] + 0.01; w.origin[1] = w.com[1] + 0.0; w.origin[2] = w.com[2] + 0.0;
        }
    }
    // ---- conservative candidate lists around every heavy atom ----------------------------
    {
        const double reach = fr.max_dist + 1e-3;
        const double reach2 = reach * reach;
        for (int p = tid; p < n; p += T::SIZE) {
            if (fr.terminal[p]) continue;
            const double* C = fr.rebuild ? &w.S[3 * ((size_t)RB_CENTRAL * n + p)] : &w.V[3 * p];
            double cx = C[0], cy = C[1], cz = C[2];
            int cnt = 0;
            for (int img = fr.rebuild ? 0 : RB_CENTRAL; img < (fr.rebuild ? 27 : RB_CENTRAL + 1); ++img) {
                const double* X;
                if (fr.rebuild) {
                    const double* b = &w.box[6 * img];
                    if (cx < b[0] - reach || cx > b[3] + reach || cy < b[1] - reach || cy > b[4] + reach ||
                        cz < b[2] - reach || cz > b[5] + reach)
                        continue;
                    X = &w.S[3 * (size_t)img * n];
                } else {
                    X = w.V;
                }
                for (int q = 0; q < n; ++q) {
                    double dx = X[3 * q] - cx, dy = X[3 * q + 1] - cy, dz = X[3 * q + 2] - cz;
                    double d2 = dx * dx + dy * dy + dz * dz;
                    if (d2 < reach2 && !(img == RB_CENTRAL && q == p)) {
                        if (cnt < RB_NB_CAP) w.nb[(size_t)p * RB_NB_CAP + cnt] = img * n + q;
                        ++cnt;
                    }
                }
            }
            w.nb_cnt[p] = cnt;
            if (cnt > RB_NB_CAP) rb_atomic_or(&w.status, RB_ST_NB_OVERFLOW);
        }
    }
    T::sync();
    // ---- molecules, one at a time ------------------------------------------------------------
    int mol_serial = 0, layer_serial = 0;
    for (;;) {
        // start: the remaining heavy atom closest to the pseudo origin (utilities.py:955-972)
        {
            double ox = w.origin[0], oy = w.origin[1], oz = w.origin[2];
            double oo = sq3(ox, oy, oz);
            double best = PW_INF;
            int bi = -1;
            for (int q = tid; q < n; q += T::SIZE) {
                if (!w.remaining[q] || fr.terminal[q]) continue;
                double d = rb_dist_sk(&w.V[3 * q], w.Vxx[q], ox, oy, oz, oo);
                if (d < best || bi < 0) { best = d; bi = q; }
            }
            w.red_v[tid] = best;
            w.red_i[tid] = bi;
            T::sync();
            if (tid == 0) {
                double b = PW_INF;
                int i0 = -1;
                for (int t = 0; t < T::SIZE; ++t) {
                    int it = w.red_i[t];
                    if (it < 0) continue;
                    double bt = w.red_v[t];
                    if (i0 < 0 || bt < b || (bt == b && it < i0)) { b = bt; i0 = it; }
                }
                w.start = i0;
            }
            T::sync();
        }
        if (w.start < 0) break;
        ++mol_serial;
        if (tid == 0) { w.work[0] = w.start; w.n_work = 1; w.n_final = 0; }
        T::sync();
        // breadth-first layers (utilities.py:982-1055)
        for (;;) {
            const int nw = w.n_work;
            if (nw == 0) break;
            ++layer_serial;
            if (tid == 0) w.n_next = 0;
            // the atoms of this layer join the molecule in list order
            for (int k = tid; k < nw; k += T::SIZE) {
                int id = w.work[k];
                w.final_[w.n_final + k] = id;
                w.stamp_final[id] = mol_serial;
            }
            T::sync();
            for (int c0 = 0; c0 < nw; c0 += RB_CHUNK) {
                const int cn = nw - c0 < RB_CHUNK ? nw - c0 : RB_CHUNK;
                for (int k = tid; k < cn; k += T::SIZE) w.seg_cnt[k] = 0;
                T::sync();
                for (int k = T::wave(); k < cn; k += T::NWAVES) rb_expand<T>(fr, w, w.work[c0 + k], k);
                T::sync();
                // every atom's hits in list-position order
                for (int k = tid; k < cn; k += T::SIZE) {
                    int m = w.seg_cnt[k] < RB_SEG_CAP ? w.seg_cnt[k] : RB_SEG_CAP;
                    long long* sgm = &w.seg[(size_t)k * RB_SEG_CAP];
                    for (int a = 1; a < m; ++a) {
                        long long v = sgm[a];
                        int b = a - 1;
                        while (b >= 0 && sgm[b] > v) { sgm[b + 1] = sgm[b]; --b; }
                        sgm[b + 1] = v;
                    }
                }
                T::sync();
                // unique(working_list_temp), then "not in final_molecule" (utilities.py:1044-1055)
                if (tid == 0) {
                    int nn = w.n_next;
                    for (int k = 0; k < cn; ++k) {
                        int m = w.seg_cnt[k] < RB_SEG_CAP ? w.seg_cnt[k] : RB_SEG_CAP;
                        const long long* sgm = &w.seg[(size_t)k * RB_SEG_CAP];
                        for (int a = 0; a < m; ++a) {
                            int id = (int)(sgm[a] & 0xffffffffll);
                            if (w.stamp_temp[id] == layer_serial) continue;
                            w.stamp_temp[id] = layer_serial;
                            if (w.stamp_final[id] == mol_serial) continue;
                            w.work_next[nn++] = id;
                        }
                    }
                    w.n_next = nn;
                }
                T::sync();
            }
            // atom_list.remove(i) for the atoms of this layer (utilities.py:1037-1039)
            for (int k = tid; k < nw; k += T::SIZE) {
                int id = w.work[k];
                if (id < n) w.remaining[id] = 0;
            }
            T::sync();
            if (tid == 0) {
                w.n_final += nw;
                w.n_work = w.n_next;
                int* t = w.work; w.work = w.work_next; w.work_next = t;
            }
            T::sync();
        }
        // ---- close the molecule ------------------------------------------------------------
        const int m = w.n_final;
        bool keep = true;
        if (fr.rebuild) {
            // centre of mass of the molecule in fractional coordinates, rounded to 8 places
            // (np.around: x * 1e8 -> rint -> / 1e8), inside [bound0, bound1) on all three axes
            for (int k = tid; k < m; k += T::SIZE) {
                int q, ax, ay, az;
                const double* P;
                rb_decode(w, n, w.final_[k], &q, &ax, &ay, &az, &P);
                w.msum[k] = fr.mass[q];
            }
            T::sync();
            for (int col = tid; col < 3; col += T::SIZE) {
                double acc = 0.0;
                for (int k = 0; k < m; ++k) {
                    int q, ax, ay, az;
                    const double* P;
                    rb_decode(w, n, w.final_[k], &q, &ax, &ay, &az, &P);
                    double t = P[col] * fr.mass[q];
                    acc = k == 0 ? t : acc + t;
                }
                w.com[col] = acc;
            }
            if (tid == SUM_THREAD) w.red_v[0] = np_sum_serial(w.msum, m);
            T::sync();
            if (tid == 0) {
                double total = w.red_v[0];
                double cf[3];
                rb_mat3(fr.lattice_inv, w.com[0] / total, w.com[1] / total, w.com[2] / total, cf);
                bool in = true;
                for (int c = 0; c < 3; ++c) {
                    double r = __builtin_rint(cf[c] * 1e8) / 1e8;
                    in = in && (r >= w.bound[0]) && (r < w.bound[1]);
                }
                w.red_i[0] = in ? 1 : 0;
            }
            T::sync();
            keep = w.red_i[0] != 0;
            T::sync();
        }
        if (keep) {
            const int base = w.n_out;
            const bool fits = base + m <= out.atoms_cap && w.n_mol < out.mols_cap;
            if (fits) {
                for (int k = tid; k < m; k += T::SIZE) {
                    int q, ax, ay, az;
                    const double* P;
                    int id = w.final_[k];
                    rb_decode(w, n, id, &q, &ax, &ay, &az, &P);
                    out.src_atom[base + k] = q;
                    out.src_image[base + k] = id < n ? (signed char)-1 : (signed char)((id - n) / n);
                    out.xyz[3 * (size_t)(base + k)] = P[0];
                    out.xyz[3 * (size_t)(base + k) + 1] = P[1];
                    out.xyz[3 * (size_t)(base + k) + 2] = P[2];
                }
            }
            T::sync();
            if (tid == 0) {
                if (fits) {
                    w.n_out = base + m;
                    w.n_mol += 1;
                    out.mol_offset[w.n_mol] = w.n_out;
                } else {
                    w.status |= (base + m > out.atoms_cap) ? RB_ST_ATOMS_OVERFLOW : RB_ST_MOLS_OVERFLOW;
                }
            }
            T::sync();
        }
    }
    if (tid == 0) {
        *out.n_mol = w.n_mol;
        *out.status = w.status;
    }
    T::sync();
}

}  // namespace pw
