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
//     then the supercell in (image, atom) order; here the pairs that can bond are found once
//     per frame around the atoms of the central image (uniform grid, single-precision screen,
//     then the reference's own test - see "candidate lists" in rebuild_frame) and the walk
//     looks an atom's list up; the hits enter the merge in that same order (cell atoms first,
//     then images) by construction of the lists;
//   * bond test = the reference's two formulas: scikit-learn's euclidean_distances for the
//     0.1 < d < max_dist pre-filter, distance() (utilities.py:80-93) against Rcov_i + Rcov_j
//     +- tol.
// Not reproduced: value-equality between an image atom and a DIFFERENT cell atom (an input
// that lists the same atom on two opposite cell faces); cells thinner than max_dist.
#pragma once
#include "pw_unit.hpp"

namespace pw {

constexpr int RB_NB_CAP = 16;      // neighbour candidates kept per heavy atom (bonded, or too close to a threshold to say)
constexpr int RB_NCELL = 2048;     // cells of the candidate grid
constexpr int RB_SEG_CAP = 32;     // bonded neighbours one atom can contribute per layer
constexpr int RB_CHUNK = 64;       // atoms of a layer expanded between two merges
constexpr int RB_LFINAL = 1024;    // atoms of the molecule being walked kept in team-shared memory
constexpr int RB_LWORK = 512;      // layer width kept in team-shared memory (wider layers: global lists)
constexpr int RB_WF_TRUNCATED = 1;  // a neighbour candidate lay outside the 3x3x3 supercell
constexpr int RB_WF_MARGINAL = 2;   // a distance within 1e-6 of a threshold of the bond test
constexpr int RB_WF_REPEAT = 4;     // the same atom met in two images (or twice by value): not a finite molecule
constexpr int RB_CENTRAL = 13;     // image (0,0,0) in the a,b,c-nested 3x3x3 enumeration
// a candidate entry: (image, atom), plus the outcome of the bond test when it cannot depend on the image
// the pair is met in (see "bond tests made once")
constexpr int RB_NB_CLEAR = 1 << 30;
constexpr int RB_NB_BONDED = 1 << 29;
constexpr int RB_NB_MASK = RB_NB_BONDED - 1;
constexpr int RB_NB_IMG_SHIFT = 24;            // entry = flags | image << 24 | atom  (n < 2^24)
constexpr int RB_NB_Q_MASK = (1 << RB_NB_IMG_SHIFT) - 1;

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
PW_HD inline void rb_atomic_or64(unsigned long long* p, unsigned long long v) {
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
    double d2 = pw_m2add(g, xx) + pp;
    return pw_sqrt(d2 > 0.0 ? d2 : 0.0);
}

// The reference's bond test between a visited atom at p and a candidate at x: scikit-learn's distance inside
// (0.1, max_dist), then distance() (utilities.py:80-93) inside (lo, hi).  *clearance = how far the comparisons
// that decided were from their thresholds.
PW_HD inline bool rb_bond(double x0, double x1, double x2, double px, double py, double pz, double pp, double lo,
                          double hi, double max_dist, double* clearance) {
    const double xx = sq3(x0, x1, x2);          // == Vxx[q] for a cell atom
    const double xv[3] = {x0, x1, x2};
    const double d = rb_dist_sk(xv, xx, px, py, pz, pp);
    const double c1 = pw_min(pw_abs(d - 0.1), pw_abs(d - max_dist));
    *clearance = c1;
    if (!(d > 0.1 && d < max_dist)) return false;
    double dx = px - x0, dy = py - x1, dz = pz - x2;
    double r2 = (dx * dx + dy * dy) + dz * dz;
    // distance(): (...) ** 0.5 on a float is libm pow, which is within one ulp of the correctly
    // rounded square root: only a comparison that close to a limit needs the libm value
    double r = pw_sqrt(r2);
    const double margin = r * 1.0e-15;
    if ((pw_abs(r - lo) <= margin || pw_abs(r - hi) <= margin) && r2 >= 2.2250738585072014e-308)
        r = pw_pow_np(r2, 0.5);
    *clearance = pw_min(c1, pw_min(pw_abs(r - lo), pw_abs(r - hi)));
    return lo < r && r < hi;
}

// scalars and small arrays every phase of the walk passes through: team-shared memory
struct RebuildShared {
    double red_v[8];
    double com[3], origin[3], bound[2];
    double box[27 * 6];
    float shift[27 * 3];           // image i = central image + shift[i] (lattice translation, single precision)
    int red_i[8];
    int n_work, n_next, n_final, start;
    double cf[3];                  // unrounded fractional centre of mass of the molecule being closed
    int wflags;                    // RB_WF_* of the walk in progress
    int off_lo[3], off_hi[3];      // image offsets it reached
    int skip, skip_cage, skip_off; // the walk from `start` is predicted (see rebuild_frame)
    int lwork[2][RB_LWORK];        // the first RB_LWORK atoms of the current and of the next layer
    int lfinal[RB_LFINAL];         // the first RB_LFINAL atoms of the molecule
};

struct RebuildWs {
    double* V;            // n x 3   value coordinates of the cell atoms
    double* Vxx;          // n
    double* S;            // 27n x 3 value coordinates of the supercell atoms (rebuild only)
    double* msum;         // 28n     masses of the molecule being closed (numpy pairwise sum)
    int* nb_cnt;          // n
    int* nb;              // n x RB_NB_CAP: image * n + atom
    int* stamp_final;     // ids: molecule serial
    int* stamp_temp;      // ids: layer serial
    int* work;            // ids
    int* work_next;       // ids
    int* final_;          // ids
    // ---- team-shared fast memory (LDS on the device), attach_fast() ----
    int* seg_cnt;         // RB_CHUNK
    long long* seg;       // RB_CHUNK x RB_SEG_CAP: position key << 32 | canonical id
    double* terms;        // 4 x term_cap: x, y, z moments and masses of a molecule being closed (same bytes as seg)
    unsigned long long* bits_final;   // ids bits: "in the current molecule"; null -> stamp_final is used
    unsigned long long* bits_temp;    // ids bits: "seen in the current layer"; null -> stamp_temp is used
    RebuildShared* sh;
    int term_cap, bit_words;
    unsigned char* remaining;   // n
    unsigned char* alias;       // n
    // what the first walk through a molecule learned (rebuild only; see "walks that can be predicted")
    // ---- the candidate scan: a uniform grid over the central image (the cell when nothing is rebuilt) ----
    float* scan;          // n x 4: x, y, z in single precision and the covalent radius of every atom - the scan only
                          // has to be conservative; an image is the central one seen from a shifted atom
    int* grid_ids;        // n: the atoms sorted by grid cell
    int* grid_start;      // RB_NCELL + 1: first position of every cell in grid_ids
    int scan_fast;        // ... the three of them are in team-shared memory (attach_fast)
    int* cage_of;         // n: serial of the first walk that visited the atom, 0 = none yet
    unsigned char* cage_off;    // n: image (0..26) in which that walk met the atom
    unsigned char* cage_ok;     // n + 1, by walk serial: the walk was clean (no truncation / marginal bond / repeat)
    int* cage_rng;        // n + 1: lowest and highest image offset per axis (2 bits each)
    double* cage_f;       // (n + 1) x 3: fractional centre of mass of that walk's molecule
    double* dorig;        // n: distance of every cell atom to the pseudo origin (the start atoms are its arg-minima)
    int* csr;             // n + 1 offsets, then the candidate lists one after the other (when team-shared memory
                          // cannot hold them)
    int status, n_mol, n_out;
    double inv_n;         // 1 / n

    PW_HD static size_t ids(int n, int rebuild) { return rebuild ? (size_t)28 * n : (size_t)n; }
    PW_HD static size_t bytes(int n, int rebuild, int team) {
        size_t id = ids(n, rebuild);
        (void)team;
        size_t d = (size_t)3 * n + n + (rebuild ? (size_t)81 * n : 0) + id + 3 * ((size_t)n + 1) + n;
        size_t i = (size_t)n + 2 * (size_t)n * RB_NB_CAP + 5 * id + 3 * ((size_t)n + 1);
        return sizeof(RebuildWs) + 64 + d * 8 + i * 4 + 4 * (size_t)n + 64 + 64 +
               scan_bytes(n) + 64;
    }
    PW_HD static size_t scan_bytes(int n) { return (size_t)n * 16 + (((size_t)n * 4 + 15) & ~(size_t)15) + ((size_t)RB_NCELL + 4) * 4; }
    // fast memory: the hit segments always, the two bit sets when `with_bits`, the scan coordinates when `with_scan`
    PW_HD static size_t fast_bytes(int n, int rebuild, bool with_bits, bool with_scan = false) {
        size_t words = (ids(n, rebuild) + 63) / 64;
        return (size_t)RB_CHUNK * RB_SEG_CAP * 8 + (size_t)RB_CHUNK * 4 + (with_bits ? 2 * words * 8 : 0) +
               ((sizeof(RebuildShared) + 15) & ~(size_t)15) + (with_bits ? 2 * (((size_t)n + 15) & ~(size_t)15) : 0) +
               (with_scan ? scan_bytes(n) : 0);
    }
    PW_HD void attach_fast(unsigned char* base, int n, int rebuild, bool with_bits, bool with_scan = false) {
        sh = (RebuildShared*)base;
        base += (sizeof(RebuildShared) + 15) & ~(size_t)15;
        seg = (long long*)base;
        terms = (double*)base;
        term_cap = RB_CHUNK * RB_SEG_CAP / 4;
        base += (size_t)RB_CHUNK * RB_SEG_CAP * 8;
        bit_words = (int)((ids(n, rebuild) + 63) / 64);
        bits_final = with_bits ? (unsigned long long*)base : nullptr;
        bits_temp = with_bits ? (unsigned long long*)base + bit_words : nullptr;
        base += with_bits ? 2 * (size_t)bit_words * 8 : 0;
        seg_cnt = (int*)base;
        base += (size_t)RB_CHUNK * 4;
        // the atom list ("remaining") is read and written at every step of the walk: with the bit sets
        // it moves into team-shared memory (carve() pointed it into the slab)
        // (and "the central image's copy of this atom is the same list item", looked up beside it)
        if (with_bits) { remaining = base; base += ((size_t)n + 15) & ~(size_t)15; alias = base; base += ((size_t)n + 15) & ~(size_t)15; }
        scan_fast = with_scan ? 1 : 0;
        if (with_scan) attach_scan(base, n);
    }
    PW_HD void attach_scan(unsigned char* base, int n) {
        scan = (float*)base; base += (size_t)n * 16;
        grid_ids = (int*)base; base += ((size_t)n * 4 + 15) & ~(size_t)15;
        grid_start = (int*)base;
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
        w->cage_f = (double*)p; p += 3 * ((size_t)n + 1) * 8;
        w->dorig = (double*)p; p += (size_t)n * 8;
        w->nb_cnt = (int*)p; p += (size_t)n * 4;
        w->nb = (int*)p; p += (size_t)n * RB_NB_CAP * 4;
        w->csr = (int*)p; p += ((size_t)n * RB_NB_CAP + n + 1) * 4;
        w->stamp_final = (int*)p; p += id * 4;
        w->stamp_temp = (int*)p; p += id * 4;
        w->work = (int*)p; p += id * 4;
        w->work_next = (int*)p; p += id * 4;
        w->final_ = (int*)p; p += id * 4;
        w->cage_of = (int*)p; p += ((size_t)n + 1) * 4;
        w->cage_rng = (int*)p; p += ((size_t)n + 1) * 4;
        w->remaining = p; p += n;
        w->alias = p; p += n;
        w->cage_off = p; p += n;
        w->cage_ok = p; p += (size_t)n + 1;
        p = (unsigned char*)(((size_t)p + 63) & ~(size_t)63);
        w->attach_scan(p, n);
        w->scan_fast = 0;
        w->inv_n = 1.0 / (double)n;
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
        // s / n without an integer division (a long instruction sequence here): the quotient of the doubles is
        // at most one off
        int img = (int)((double)s * w.inv_n);
        if ((img + 1) * n <= s) ++img;
        else if (img * n > s) --img;
        *q = s - img * n;
        *ax = img / 9 - 1; *ay = (img / 3) % 3 - 1; *az = img % 3 - 1;
        *pos = &w.S[3 * (size_t)s];
    }
}

// one wave expands one atom of the current layer: lanes over its candidate list
template <class T>
PW_HD inline void rb_expand(const RebuildFrame& fr, const RebuildWs& w, const int* nb_off, const int* nb_ent,
                            int* status, int* wflags, int id, int slot, long long* xprof = nullptr) {
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    long long xt = wall_clock64();
#define RB_XT(k) do { if (xprof && T::lane() == 0) { long long t_ = wall_clock64(); xprof[k] += t_ - xt; xt = t_; } } while (0)
#else
#define RB_XT(k) do { } while (0)
#endif
    const int n = fr.n;
    PW_ASSUME_LDS(w.seg);
    PW_ASSUME_LDS(w.seg_cnt);
    int q0, ax, ay, az;
    const double* P;
    rb_decode(w, n, id, &q0, &ax, &ay, &az, &P);
    // the atom's list (empty for the elements the walk does not expand: their lists were never built)
    const int beg = nb_off[q0], cnt = nb_off[q0 + 1] - beg;
    const int first = T::lane() < cnt ? nb_ent[beg + T::lane()] : 0;
    RB_XT(0);
    if (cnt == 0) return;
    for (int e = T::lane(); e < cnt; e += T::WSIZE) {
        int packed = e == T::lane() ? first : nb_ent[beg + e];
        // bond tests made once: the candidate scan has already tested this pair in the central image; when
        // no comparison came near a threshold the outcome is the same in whichever image the walk meets it
        const bool clear = (packed & RB_NB_CLEAR) != 0, bonded = (packed & RB_NB_BONDED) != 0;
        const int dimg = (packed >> RB_NB_IMG_SHIFT) & 31;
        const int q = packed & RB_NB_Q_MASK;
        int bx = ax + dimg / 9 - 1, by = ay + (dimg / 3) % 3 - 1, bz = az + dimg % 3 - 1;
        bool central = bx == 0 && by == 0 && bz == 0;
        const bool outside = bx < -1 || bx > 1 || by < -1 || by > 1 || bz < -1 || bz > 1;
        if (fr.rebuild && outside) rb_atomic_or(wflags, RB_WF_TRUNCATED);
        if (clear && !bonded) continue;
        // a candidate is tested as a remaining cell atom (utilities.py:996-1013) and / or as a
        // supercell atom that is not, by value, in the remaining atom list (:1014-1036); nearly
        // always exactly one of the two applies, so the lanes pick theirs first and share one
        // pass through the distance code (a second pass only if some lane needs both)
        const unsigned char rem = central ? w.remaining[q] : (unsigned char)0;
        const bool same_item = central && w.alias[q];
        const bool do0 = central && rem;
        const bool do1 = fr.rebuild && !outside && !(same_item && rem);
        const int s1 = ((bx + 1) * 9 + (by + 1) * 3 + (bz + 1)) * n + q;
        RB_XT(1);
        for (int pass = 0; pass < 2; ++pass) {
            const bool as_cell = pass == 0 ? do0 : false;
            const bool as_image = pass == 0 ? (!do0 && do1) : (do0 && do1);
            if (pass == 1 && !T::wave_any(as_image)) break;
            if (!as_cell && !as_image) continue;
            const long long key = as_cell ? (((long long)q << 32) | (unsigned)q)
                                          : (((long long)(n + s1) << 32) | (unsigned)(same_item ? q : n + s1));
            if (!clear) {
                // (rare: a comparison of this pair's bond test is too close to its threshold to be made once)
                const double* X = as_cell ? &w.V[3 * q] : &w.S[3 * (size_t)s1];
                const double x0 = X[0], x1 = X[1], x2 = X[2];
                const double px = P[0], py = P[1], pz = P[2];
                const double pp = sq3(px, py, pz);
                RB_XT(2);
                const double rc = fr.cov[q0] + fr.cov[q];
                double clearance;
                const bool hit = rb_bond(x0, x1, x2, px, py, pz, pp, rc - fr.tol, rc + fr.tol, fr.max_dist, &clearance);
                if (clearance < 1e-6) rb_atomic_or(wflags, RB_WF_MARGINAL);
                if (!hit) continue;
            }
            int k = rb_atomic_add(&w.seg_cnt[slot], 1);
            if (k < RB_SEG_CAP) w.seg[(size_t)slot * RB_SEG_CAP + k] = key;
            else rb_atomic_or(status, RB_ST_SEG_OVERFLOW);
        }
        RB_XT(3);
    }
    RB_XT(4);
}

#if defined(__HIP_DEVICE_COMPILE__)
// One wavefront walks one molecule.  The layers of a molecule are a few atoms wide, and a layer handled by the
// whole team costs two team barriers whatever its width; here the four phases of a layer - the layer's atoms
// join the molecule, their lists are expanded, the hits are merged in list order into the next layer, the
// layer's atoms leave the atom list (utilities.py:982-1055) - run in one wave, ordered by nothing more than
// the wave's own LDS queue:
//   * expansion: the list entries of up to 64 atoms of the layer are dealt one per lane (prefix sum of the list
//     lengths; at most 64 entries per pass).  An entry's hits need no sorting: a list is in (image, atom) order,
//     which is the order of the reference's supercell scan whatever image the atom sits in, and the remaining
//     cell atoms - scanned first by the reference - are one run of it (the central image), so the hits of an
//     atom are its cell hits followed by its image hits, each in list order: positions by ballots;
//   * merge: one hit per lane - first occurrence by lane-to-lane comparison, "seen in this layer" and "in the
//     molecule" from the bit sets, positions in the next layer from a ballot.
// Returns the number of layers.  Needs the visit bit sets (and the atom list) in team-shared memory.
template <class T, class LISTP>
__device__ inline int rb_wave_walk(const RebuildFrame& fr, const RebuildWs& WS, LISTP nb_off, LISTP nb_ent, int start,
                                   long long* wprof = nullptr) {
#if defined(PW_RB_PROFILE)
    long long wt = wall_clock64();
#define RB_WT(k) do { if (wprof && T::lane() == 0) { long long t_ = wall_clock64(); wprof[k] += t_ - wt; wt = t_; } } while (0)
#else
#define RB_WT(k) do { } while (0)
#endif
    PW_LDS RebuildShared& sh = *(PW_LDS RebuildShared*)WS.sh;
    PW_LDS unsigned long long* const bits_final = (PW_LDS unsigned long long*)WS.bits_final;
    PW_LDS unsigned long long* const bits_temp = (PW_LDS unsigned long long*)WS.bits_temp;
    PW_LDS int* const hits = (PW_LDS int*)WS.seg;
    PW_LDS unsigned char* const remaining = (PW_LDS unsigned char*)WS.remaining;
    const PW_LDS unsigned char* const alias = (const PW_LDS unsigned char*)WS.alias;
    int* work = WS.work;
    int* work_next = WS.work_next;
    const int n = fr.n, lane = T::lane();
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cur = 0, nw = 1, nf = 0, layers = 0;
    if (lane == 0) sh.lwork[0][0] = start;
    T::wave_sync();
    while (nw > 0) {
        const PW_LDS int* wl = sh.lwork[cur];
        PW_LDS int* wn = sh.lwork[cur ^ 1];
        auto LW = [&](int k) { return k < RB_LWORK ? wl[k] : work[k]; };
        for (int k = lane; k < nw; k += 64) {
            const int id = LW(k);
            if (nf + k < RB_LFINAL) sh.lfinal[nf + k] = id;
            else WS.final_[nf + k] = id;
            atomicOr((unsigned long long*)&bits_final[id >> 6], 1ull << (id & 63));
        }
        T::wave_sync();
        RB_WT(0);
        int nn = 0;
        // "seen in this layer": first occurrences inside a merge round are found lane to lane; the bit set is only
        // needed - and only then cleared - when a layer takes more than one round
        bool temp_live = false;
        for (int a0 = 0; a0 < nw;) {
            // lane k: atom a0 + k of the layer
            const int ka = a0 + lane;
            const bool va = ka < nw;
            int id0 = 0, q0 = 0, ax = 0, ay = 0, az = 0, beg = 0, cnt = 0;
            if (va) {
                const double* P;
                id0 = LW(ka);
                rb_decode(WS, n, id0, &q0, &ax, &ay, &az, &P);
                beg = nb_off[q0];
                cnt = nb_off[q0 + 1] - beg;
            }
            RB_WT(1);
            const int incl = T::incl_scan_i(cnt);
            // the atoms whose entries fit this pass (a list has at most RB_NB_CAP entries: never none)
            const int na = __popcll(__ballot(va && incl <= 64));
            const int E = __shfl(incl, na - 1);
            // lane l: entry l of the pass; its atom
            int j = 0;
            for (int k = 0; k < na; ++k) j += (__builtin_amdgcn_readlane(incl, k) <= lane) ? 1 : 0;
            const bool ve = lane < E;
            const int jj = ve ? j : 0;
            const int j_incl = __shfl(incl, jj), j_cnt = __shfl(cnt, jj), j_beg = __shfl(beg, jj);
            const int j_q0 = __shfl(q0, jj), j_id = __shfl(id0, jj);
            const int j_a = __shfl((ax + 1) | ((ay + 1) << 2) | ((az + 1) << 4), jj);
            const int j_excl = j_incl - j_cnt;
            RB_WT(2);
            bool cell_hit = false, img_hit = false;
            int cell_id = 0, img_id = 0;
            if (ve) {
                const int packed = nb_ent[j_beg + (lane - j_excl)];
                const bool clear = (packed & RB_NB_CLEAR) != 0, bonded = (packed & RB_NB_BONDED) != 0;
                const int dimg = (packed >> RB_NB_IMG_SHIFT) & 31;
                const int q = packed & RB_NB_Q_MASK;
                const int bx = (j_a & 3) - 1 + dimg / 9 - 1, by = ((j_a >> 2) & 3) - 1 + (dimg / 3) % 3 - 1,
                          bz = ((j_a >> 4) & 3) - 1 + dimg % 3 - 1;
                const bool central = bx == 0 && by == 0 && bz == 0;
                const bool outside = bx < -1 || bx > 1 || by < -1 || by > 1 || bz < -1 || bz > 1;
                if (fr.rebuild && outside) atomicOr((int*)&sh.wflags, RB_WF_TRUNCATED);
                // as a remaining cell atom (utilities.py:996-1013) and / or as a supercell atom that is not, by
                // value, in the remaining atom list (:1014-1036)
                const unsigned char rem = central ? remaining[q] : (unsigned char)0;
                const bool same_item = central && alias[q];
                const bool do0 = central && rem;
                const bool do1 = fr.rebuild && !outside && !(same_item && rem);
                const int s1 = ((bx + 1) * 9 + (by + 1) * 3 + (bz + 1)) * n + q;
                bool t0 = bonded, t1 = bonded;
                if (!clear && (do0 || do1)) {
                    // (rare: a comparison of this pair's bond test is too close to its threshold to be made once)
                    int qq, a1, a2, a3;
                    const double* P;
                    rb_decode(WS, n, j_id, &qq, &a1, &a2, &a3, &P);
                    const double px = P[0], py = P[1], pz = P[2];
                    const double pp = sq3(px, py, pz);
                    const double rc = fr.cov[j_q0] + fr.cov[q];
                    double clearance;
                    if (do0) {
                        const double* X = &WS.V[3 * q];
                        t0 = rb_bond(X[0], X[1], X[2], px, py, pz, pp, rc - fr.tol, rc + fr.tol, fr.max_dist, &clearance);
                        if (clearance < 1e-6) atomicOr((int*)&sh.wflags, RB_WF_MARGINAL);
                    }
                    if (do1) {
                        const double* X = &WS.S[3 * (size_t)s1];
                        t1 = rb_bond(X[0], X[1], X[2], px, py, pz, pp, rc - fr.tol, rc + fr.tol, fr.max_dist, &clearance);
                        if (clearance < 1e-6) atomicOr((int*)&sh.wflags, RB_WF_MARGINAL);
                    }
                }
                cell_hit = do0 && t0;
                img_hit = do1 && t1;
                cell_id = q;
                img_id = same_item ? q : n + s1;
            }
            const unsigned long long C = __ballot(cell_hit), I = __ballot(img_hit);
            const int H = __popcll(C) + __popcll(I);
            RB_WT(3);
            if (H) {
                const bool use_temp = !(a0 == 0 && na >= nw && H <= 64);
                if (use_temp && !temp_live) {
                    for (int i = lane; i < WS.bit_words; i += 64) bits_temp[i] = 0;
                    temp_live = true;
                }
                const unsigned long long below = j_excl == 0 ? 0ull : (~0ull >> (64 - j_excl));
                const unsigned long long upto = j_incl >= 64 ? ~0ull : ((1ull << j_incl) - 1ull);
                const unsigned long long segm = upto & ~below;
                const int base = __popcll(C & below) + __popcll(I & below);
                const int ncell = __popcll(C & segm);
                if (cell_hit) hits[base + __popcll(C & segm & lt)] = cell_id;
                if (img_hit) hits[base + ncell + __popcll(I & segm & lt)] = img_id;
                T::wave_sync();
                // unique(working_list_temp), then "not in final_molecule" (utilities.py:1044-1055)
                for (int h0 = 0; h0 < H; h0 += 64) {
                    const int hn = H - h0 < 64 ? H - h0 : 64;
                    const bool act = lane < hn;
                    const int id = act ? hits[h0 + lane] : -1 - lane;       // (distinct dummies for idle lanes)
                    bool dup = false;
                    for (int t = 0; t < hn; ++t) { const int idt = __builtin_amdgcn_readlane(id, t); dup = dup || (t < lane && idt == id); }
                    bool keep = false;
                    if (act && !dup) {
                        const unsigned long long bit = 1ull << (id & 63);
                        const bool fresh = use_temp ? !(bits_temp[id >> 6] & bit) : true;
                        keep = fresh && !(bits_final[id >> 6] & bit);
                        if (use_temp && fresh) atomicOr((unsigned long long*)&bits_temp[id >> 6], bit);
                    }
                    const unsigned long long bal = __ballot(keep);
                    const int pos = nn + __popcll(bal & lt);
                    if (keep) {
                        if (pos < RB_LWORK) wn[pos] = id;
                        else work_next[pos] = id;
                    }
                    nn += __popcll(bal);
                    T::wave_sync();
                }
            }
            RB_WT(4);
            a0 += na;
        }
        // atom_list.remove(i) for the atoms of this layer, after the last of them has been expanded
        for (int k = lane; k < nw; k += 64) {
            const int id = LW(k);
            if (id < n) remaining[id] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // (layers wider than RB_LWORK: tails in global lists)
        T::wave_sync();
        nf += nw;
        nw = nn;
        cur ^= 1;
        { int* t = work; work = work_next; work_next = t; }
        ++layers;
        RB_WT(5);
    }
    if (lane == 0) { sh.n_final = nf; sh.n_work = 0; sh.n_next = 0; }
    return layers;
}
#endif

// -DPW_RB_PROFILE: team 0 prints the wall time of its phases per frame (tests/tools/rebuild_profile.py)
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define RB_TICK(k) do { if (T::tid() == 0) { long long t_ = wall_clock64(); rb_prof[k] += t_ - rb_last; rb_last = t_; } } while (0)
#else
#define RB_TICK(k) do { } while (0)
#endif
template <class T>
PW_HD inline void rebuild_frame(const RebuildFrame& fr, RebuildWs& w, const RebuildOut& out) {
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    long long rb_prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long rb_last = wall_clock64();
    int rb_layers = 0, rb_mols = 0, rb_rounds = 0, rb_heavy = 0, rb_imgs = 0, rb_skipped = 0;
    long long rb_xp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // the array pointers of the workspace header, copied once: the header itself lives in global
    // memory, and reading a pointer through it costs a dependent load after every store
    const RebuildWs WS = w;
    // team-shared arrays through address-space-qualified pointers (ds_read / ds_write, not flat)
    PW_LDS RebuildShared& sh = *(PW_LDS RebuildShared*)WS.sh;
    PW_LDS long long* const seg = (PW_LDS long long*)WS.seg;
    PW_LDS int* const seg_cnt = (PW_LDS int*)WS.seg_cnt;
    PW_LDS double* const terms = (PW_LDS double*)WS.terms;
    PW_LDS unsigned long long* const bits_final = (PW_LDS unsigned long long*)WS.bits_final;
    PW_LDS unsigned long long* const bits_temp = (PW_LDS unsigned long long*)WS.bits_temp;
    int* work = WS.work;
    int* work_next = WS.work_next;
    const int n = fr.n;
    const int n_ids = (int)RebuildWs::ids(n, fr.rebuild);
    const int tid = T::tid();
    static_assert(T::NWAVES <= 4, "red_v[4 + wave] carries the coordinate maxima");
    if (tid == 0) {
        w.status = 0; w.n_mol = 0; w.n_out = 0;
        out.mol_offset[0] = 0;
    }
    // ---- system centre of mass (utilities.py:127-148 on the unrounded input) -----------
    // numpy adds the rows one after the other: three threads, one chain of additions each.  The products are
    // staged in team-shared memory by everybody first (the scan arrays are not written yet) so that the chains
    // do not wait for global memory.
    constexpr int SUM_THREAD = T::SIZE > 3 ? 3 : 0;
    const bool com_staged = WS.scan_fast && n <= RB_CHUNK * RB_SEG_CAP && (size_t)24 * n <= RebuildWs::scan_bytes(n);
    if (com_staged) {
        PW_LDS double* t3 = (PW_LDS double*)WS.scan;
        PW_LDS double* tm = (PW_LDS double*)WS.seg;
        for (int i = tid; i < n; i += T::SIZE) {
            const double m = fr.mass[i];
            t3[i] = fr.xyz[3 * i] * m; t3[(size_t)n + i] = fr.xyz[3 * i + 1] * m; t3[2 * (size_t)n + i] = fr.xyz[3 * i + 2] * m;
            tm[i] = m;
        }
        T::sync();
        for (int col = tid; col < 3; col += T::SIZE)
            sh.com[col] = seq_sum_blocked(n, [&](int r) { return t3[(size_t)col * n + r]; });
        if (tid == SUM_THREAD) sh.red_v[0] = np_sum_lean((const double*)WS.seg, n);
        T::sync();
    }
    // ---- value coordinates ------------------------------------------------------------
    double vmax = 0.0;                                 // largest coordinate of the cell (for clear_min, below)
    for (int i = tid; i < n; i += T::SIZE) {
        double x = rb_round8(fr.xyz[3 * i]), y = rb_round8(fr.xyz[3 * i + 1]), z = rb_round8(fr.xyz[3 * i + 2]);
        vmax = pw_max(vmax, pw_max(pw_abs(x), pw_max(pw_abs(y), pw_abs(z))));
        WS.V[3 * i] = x; WS.V[3 * i + 1] = y; WS.V[3 * i + 2] = z;
        WS.Vxx[i] = sq3(x, y, z);
        if (!fr.rebuild) { WS.scan[4 * i] = (float)x; WS.scan[4 * i + 1] = (float)y; WS.scan[4 * i + 2] = (float)z; WS.scan[4 * i + 3] = (float)fr.cov[i]; }
        WS.remaining[i] = 1;
        WS.alias[i] = 0;
        WS.nb_cnt[i] = 0;
        WS.cage_of[i] = 0;
    }
    {
        const double wm = -T::wave_min(-vmax);
        if (T::lane() == 0) sh.red_v[4 + T::wave()] = wm;
    }
    const bool use_bits = WS.bits_final != nullptr;
    if (use_bits) {
        for (int i = tid; i < WS.bit_words; i += T::SIZE) { bits_final[i] = 0; bits_temp[i] = 0; }
    } else {
        for (int i = tid; i < n_ids; i += T::SIZE) { WS.stamp_final[i] = 0; WS.stamp_temp[i] = 0; }
    }
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
                WS.S[3 * s] = rb_round8(c[0]); WS.S[3 * s + 1] = rb_round8(c[1]); WS.S[3 * s + 2] = rb_round8(c[2]);
                if (img == RB_CENTRAL) { WS.scan[4 * i] = (float)c[0]; WS.scan[4 * i + 1] = (float)c[1]; WS.scan[4 * i + 2] = (float)c[2]; WS.scan[4 * i + 3] = (float)fr.cov[i]; }
            }
            size_t s0 = (size_t)RB_CENTRAL * n + i;
            WS.alias[i] = (WS.S[3 * s0] == WS.V[3 * i] && WS.S[3 * s0 + 1] == WS.V[3 * i + 1] &&
                          WS.S[3 * s0 + 2] == WS.V[3 * i + 2]) ? 1 : 0;
        }
    }
    T::sync();
    RB_TICK(0);
    if (!com_staged) {
        for (int col = tid; col < 3; col += T::SIZE)
            sh.com[col] = seq_sum_blocked(n, [&](int r) { return fr.xyz[3 * r + col] * fr.mass[r]; });
        if (tid == SUM_THREAD) sh.red_v[0] = np_sum_lean(fr.mass, n);
    }
    T::sync();
    if (tid == 0) {
        double total = sh.red_v[0];
        for (int c = 0; c < 3; ++c) sh.com[c] = sh.com[c] / total;
        if (fr.periodic) {
            // origin skewed by 0.01 along x; pseudo origin at fractional (0.26, 0.25, 0.25)
            // (utilities.py:889-899); <-0.5, 0.5> cell when the system COM is at the origin
            rb_mat3(fr.lattice, 0.26, 0.25, 0.25, (double*)sh.origin);
            bool centred = pw_abs(sh.com[0] - 0.01) <= 1.0 + 1e-5 * 0.01 && pw_abs(sh.com[1]) <= 1.0 &&
                           pw_abs(sh.com[2]) <= 1.0;
            sh.bound[0] = centred ? -0.5 : 0.0;
            sh.bound[1] = centred ? 0.5 : 1.0;
            // candidate lists assume a bond never spans two images
            double h[3];
            for (int c = 0; c < 3; ++c) h[c] = pw_abs(fr.lattice[3 * c + c]);
            if (fr.rebuild && (h[0] < fr.max_dist || h[1] < fr.max_dist || h[2] < fr.max_dist))
                w.status |= RB_ST_THIN_CELL;
        } else {
            sh.origin[0] = sh.com[0] + 0.01; sh.origin[1] = sh.com[1] + 0.0; sh.origin[2] = sh.com[2] + 0.0;
        }
    }
    RB_TICK(1);
    // ---- candidate lists around every heavy atom ------------------------------------------------
    // Which pairs of atoms the walk has to test is a property of the frame, not of the walk: the value
    // coordinates of an image are the central image's plus a lattice translation (and 1e-8 of rounding), so the
    // pair "atom p in image a, atom q in image a + d" is the pair "p in the central image, q in image d".  The
    // lists are therefore built once, around the atoms of the central image:
    //   * a uniform grid over the central image (cells no smaller than the reach of a bond, atoms sorted by
    //     cell) in single precision: an image is the central one seen from the atom shifted back by the
    //     image's translation, and only the 27 cells around that point are looked at - one atom per lane;
    //   * a single-precision screen with the pair's own limits (covalent radii + tolerance, max_dist), 2e-3 and
    //     the rounding of the coordinates to spare: what fails it fails the bond test in every image;
    //   * bond tests made once: the survivors are tested with the reference's arithmetic on the value
    //     coordinates.  When no comparison of the test comes within `clear_min` of its threshold the outcome -
    //     and the absence of a marginal comparison - is the same in whichever image the walk meets the pair:
    //     the bonded ones are marked and the walk appends them without touching a coordinate, the others are
    //     dropped (an unbonded candidate outside the supercell cuts nothing off a walk); a pair near a
    //     threshold stays unmarked and is tested where it is met.
    // A list is kept in (image, atom) order by insertion (a handful of entries).
    {
        static_assert((size_t)RB_NB_CAP * T::SIZE * 4 <= (size_t)RB_CHUNK * RB_SEG_CAP * 8, "the lists are staged in the hit segments");
        if (fr.rebuild) {
            for (int i = tid; i < 27 * 3; i += T::SIZE) {
                const int img = i / 3, c = i % 3;
                sh.shift[i] = (float)(WS.S[3 * ((size_t)img * n) + c] - WS.S[3 * ((size_t)RB_CENTRAL * n) + c]);
            }
        }
        double amax = 0.0, lsum = 0.0;
        for (int t = 0; t < T::NWAVES; ++t) amax = pw_max(amax, sh.red_v[4 + t]);
        if (fr.rebuild) for (int i = 0; i < 9; ++i) lsum += pw_abs(fr.lattice[i]);
        const double ext = amax + lsum + 1.0;       // no coordinate of the supercell is larger
        // comparisons this far from their thresholds come out the same in every image: 1e-6 is where a
        // comparison counts as marginal, 1e-7 covers the rounding of the value coordinates, the last term the
        // rounding of scikit-learn's formula at the largest coordinates of the supercell
        const double clear_min = 1.1e-6 + 6e-14 * ext * ext;
        // single precision moves a coordinate by at most 6e-8 of its size (the atom's, the translation's, the
        // shifted atom's): the slack of the screen grows with the supercell
        const float slack = (float)(2e-3 + 1e-6 * ext);
        const float reach = (float)fr.max_dist + slack;
        // the grid: bounding box of the central image, cells of side >= reach, at most RB_NCELL of them
        auto build = [&](auto X, auto ids, auto cs) __attribute__((always_inline)) {
            float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            for (int q = tid; q < n; q += T::SIZE)
                for (int c = 0; c < 3; ++c) {
                    const float v = X[4 * q + c];
                    lo[c] = v < lo[c] ? v : lo[c]; hi[c] = v > hi[c] ? v : hi[c];
                }
            for (int c = 0; c < 3; ++c) {
                const double l = T::wave_min((double)lo[c]), h = -T::wave_min(-(double)hi[c]);
                if (T::lane() == 0) { sh.box[6 * T::wave() + c] = l; sh.box[6 * T::wave() + 3 + c] = h; }
            }
            for (int c = tid; c <= RB_NCELL + 1; c += T::SIZE) cs[c] = 0;
            T::sync();
            float g0[3], ginv, gh = reach + slack;
            int gn[3];
            for (int c = 0; c < 3; ++c) {
                double l = sh.box[c], h = sh.box[3 + c];
                for (int t = 1; t < T::NWAVES; ++t) { l = pw_min(l, sh.box[6 * t + c]); h = pw_max(h, sh.box[6 * t + 3 + c]); }
                g0[c] = (float)l; hi[c] = (float)h;
            }
            long cells;
            for (;;) {
                ginv = 1.0f / gh;
                cells = 1;
                for (int c = 0; c < 3; ++c) {
                    const float w_ = (hi[c] - g0[c]) * ginv;
                    gn[c] = w_ < 1.0e6f ? (int)w_ + 1 : 1000000;
                    cells *= gn[c];
                }
                if (cells <= RB_NCELL) break;
                gh *= 1.26f;
            }
            auto cell_of = [&](float v, int c) { int i = (int)((v - g0[c]) * ginv); return i < 0 ? 0 : (i >= gn[c] ? gn[c] - 1 : i); };
            // atoms per cell, first positions (cs[c + 1] holds the start of cell c until the atoms are dealt,
            // its end - the start of cell c + 1 - afterwards), atoms by cell
            for (int q = tid; q < n; q += T::SIZE) {
                const int c = (cell_of(X[4 * q + 2], 2) * gn[1] + cell_of(X[4 * q + 1], 1)) * gn[0] + cell_of(X[4 * q], 0);
                rb_atomic_add((int*)&cs[c + 2], 1);
            }
            T::sync();
            if (T::wave() == 0) {
                // inclusive scan of cs[2..]: cs[c + 2] = atoms in cells 0..c = start of cell c + 1
                int run = 0;
                for (int b0 = 0; b0 < (int)cells; b0 += T::WSIZE) {
                    const int c = b0 + T::lane() + 2;
                    int v = c <= (int)cells + 1 ? cs[c] : 0, incl = v;
                    for (int d = 1; d < T::WSIZE; d <<= 1) { int t = T::shfl_up_i(incl, d); if (T::lane() >= d) incl += t; }
                    if (c <= (int)cells + 1) cs[c] = run + incl;
                    run += T::bcast_i(incl, T::WSIZE - 1);
                }
            }
            T::sync();
            for (int q = tid; q < n; q += T::SIZE) {
                const int c = (cell_of(X[4 * q + 2], 2) * gn[1] + cell_of(X[4 * q + 1], 1)) * gn[0] + cell_of(X[4 * q], 0);
                const int pos = rb_atomic_add((int*)&cs[c + 1], 1);
                ids[pos] = q;
            }
            T::sync();
            // ---- one atom per lane ----
            auto stage = [&](int e) -> decltype(auto) { return ((PW_LDS int*)seg)[e * T::SIZE + tid]; };
            for (int p0 = 0; p0 < n; p0 += T::SIZE) {
                const int p = p0 + tid;
                const bool heavy = p < n && !fr.terminal[p];
                int cnt = 0;
                double cx = 0.0, cy = 0.0, cz = 0.0, rp = 0.0;
                if (heavy) {
                    const double* C = fr.rebuild ? &WS.S[3 * ((size_t)RB_CENTRAL * n + p)] : &WS.V[3 * p];
                    cx = C[0]; cy = C[1]; cz = C[2]; rp = fr.cov[p];
                }
                const float fcx = (float)cx, fcy = (float)cy, fcz = (float)cz, frp = (float)rp;
                const double pp = sq3(cx, cy, cz);
                for (int img = 0; img < (fr.rebuild ? 27 : 1); ++img) {
                    const int im = fr.rebuild ? img : RB_CENTRAL;
                    // the atom as the central image sees it from this image
                    const float r[3] = {fcx - (fr.rebuild ? sh.shift[3 * img] : 0.0f), fcy - (fr.rebuild ? sh.shift[3 * img + 1] : 0.0f),
                                        fcz - (fr.rebuild ? sh.shift[3 * img + 2] : 0.0f)};
                    int c0[3], c1[3];
                    bool act = heavy;
                    for (int c = 0; c < 3; ++c) {
                        const float f = (r[c] - g0[c]) * ginv;
                        // (a point a whole cell or more outside the grid has no neighbour in it)
                        if (!(f > -1.0f && f < (float)gn[c] + 1.0f)) { act = false; c0[c] = 0; c1[c] = -1; continue; }
                        const int i = (int)(f + 1.0f) - 1;              // floor for f > -1
                        c0[c] = i - 1 < 0 ? 0 : i - 1;
                        c1[c] = i + 1 >= gn[c] ? gn[c] - 1 : i + 1;
                    }
                    if (!T::wave_any(act)) continue;
                    if (!act) continue;
                    for (int iz = c0[2]; iz <= c1[2]; ++iz)
                        for (int iy = c0[1]; iy <= c1[1]; ++iy) {
                            const int row = (iz * gn[1] + iy) * gn[0];
                            const int k0 = cs[row + c0[0]], k1 = cs[row + c1[0] + 1];
                            for (int k = k0; k < k1; ++k) {
                                const int q = ids[k];
                                const float qx = X[4 * q], qy = X[4 * q + 1], qz = X[4 * q + 2], qr = X[4 * q + 3];
                                const float dx = qx - r[0], dy = qy - r[1], dz = qz - r[2];
                                const float d2 = dx * dx + dy * dy + dz * dz;
                                // the screen: inside (0.1, max_dist) and inside (Rcov sum -+ tol), generously
                                const float hi_f = frp + qr + (float)fr.tol + slack;
                                float lo_f = frp + qr - (float)fr.tol - slack;
                                lo_f = lo_f > 0.1f - slack ? lo_f : 0.1f - slack;
                                lo_f = lo_f > 0.0f ? lo_f : 0.0f;
                                const float up = hi_f < reach ? hi_f : reach;
                                if (!(d2 < up * up && d2 > lo_f * lo_f) || (im == RB_CENTRAL && q == p)) continue;
                                if (cnt < RB_NB_CAP) stage(cnt) = (im << RB_NB_IMG_SHIFT) | q;
                                ++cnt;
                            }
                        }
                }
                // the reference's test on the value coordinates of the survivors (entry e of every lane at the
                // same time: the loads of a round are in flight together), kept in (image, atom) order
                const int m = cnt < RB_NB_CAP ? cnt : RB_NB_CAP;
                int kept = 0;
                for (int e = 0; e < m; ++e) {
                    const int key = stage(e);
                    const int img = key >> RB_NB_IMG_SHIFT, q = key & RB_NB_Q_MASK;
                    const double* Xc = fr.rebuild ? &WS.S[3 * ((size_t)img * n + q)] : &WS.V[3 * q];
                    const double rc = rp + fr.cov[q];
                    double clearance;
                    const bool hit = rb_bond(Xc[0], Xc[1], Xc[2], cx, cy, cz, pp, rc - fr.tol, rc + fr.tol, fr.max_dist,
                                             &clearance);
                    const bool clear = clearance >= clear_min;
                    if (!hit && clear) continue;
                    const int val = key | (clear ? RB_NB_CLEAR : 0) | (hit ? RB_NB_BONDED : 0);
                    int k = kept;
                    while (k > 0 && (stage(k - 1) & RB_NB_MASK) > key) { stage(k) = stage(k - 1); --k; }
                    stage(k) = val;
                    ++kept;
                }
                if (heavy) {
                    for (int e = 0; e < kept; ++e) WS.nb[(size_t)p * RB_NB_CAP + e] = stage(e);
                    WS.nb_cnt[p] = kept;
                    if (cnt > RB_NB_CAP) rb_atomic_or(&w.status, RB_ST_NB_OVERFLOW);
                }
            }
        };
        if (WS.scan_fast) build((const PW_LDS float*)WS.scan, (PW_LDS int*)WS.grid_ids, (PW_LDS int*)WS.grid_start);
        else build((const float*)WS.scan, WS.grid_ids, WS.grid_start);
    }
    T::sync();
    // ---- what the walk reads at every step, side by side in the memory the scan no longer needs ------------
    // the lists one after the other behind an offset per atom, and the distance of every heavy atom to the
    // pseudo origin (every start atom is the remaining heavy atom closest to it, utilities.py:955-972; the
    // distances do not change during the frame).  Team-shared memory when the scan arrays were there and the
    // lists fit, the team's slab otherwise.
    const int* nb_off;
    const int* nb_ent;
    const double* dorig;
    bool lists_fast;
    {
        const int per = (n + T::SIZE - 1) / T::SIZE;
        const int qa = tid * per < n ? tid * per : n, qb = qa + per < n ? qa + per : n;
        int sum = 0;
        for (int q = qa; q < qb; ++q) sum += WS.nb_cnt[q];
        int incl = sum;
        for (int d = 1; d < T::WSIZE; d <<= 1) { int t = T::shfl_up_i(incl, d); if (T::lane() >= d) incl += t; }
        if (T::lane() == T::WSIZE - 1) sh.red_i[T::wave()] = incl;
        T::sync();
        int run = incl - sum, total = 0;
        for (int t = 0; t < T::NWAVES; ++t) { if (t < T::wave()) run += sh.red_i[t]; total += sh.red_i[t]; }
        const size_t fast_ints = RebuildWs::scan_bytes(n) / 4;
        const bool fast = WS.scan_fast && 2 * (size_t)n + (size_t)n + 1 + (size_t)total <= fast_ints;
        double* dd = fast ? (double*)WS.scan : WS.dorig;
        int* off = fast ? (int*)WS.scan + 2 * (size_t)n : WS.csr;
        int* ent = off + n + 1;
        for (int q = qa; q < qb; ++q) {
            const int c = WS.nb_cnt[q];
            off[q] = run;
            for (int e = 0; e < c; ++e) ent[run + e] = WS.nb[(size_t)q * RB_NB_CAP + e];
            run += c;
        }
        if (tid == 0) off[n] = total;
        const double ox = sh.origin[0], oy = sh.origin[1], oz = sh.origin[2];
        const double oo = sq3(ox, oy, oz);
        for (int q = tid; q < n; q += T::SIZE)
            dd[q] = fr.terminal[q] ? PW_INF : rb_dist_sk(&WS.V[3 * q], WS.Vxx[q], ox, oy, oz, oo);
        nb_off = off; nb_ent = ent; dorig = dd; lists_fast = fast;
    }
    T::sync();
    RB_TICK(2);
    // ---- molecules, one at a time ------------------------------------------------------------
    int mol_serial = 0, layer_serial = 0;
    for (;;) {
        // start: the remaining heavy atom closest to the pseudo origin (utilities.py:955-972)
        {
            double best = PW_INF;
            int bi = -1;
            for (int q = tid; q < n; q += T::SIZE) {
                if (!WS.remaining[q]) continue;
                double d = dorig[q];
                if (!(d < PW_INF)) continue;           // (an element the walk does not start from)
                if (d < best || bi < 0) { best = d; bi = q; }
            }
            if (bi < 0) { best = PW_INF; bi = 0x7fffffff; }
            T::wave_argmin(best, bi);            // smallest distance, smallest index on ties
            if (T::lane() == 0) { sh.red_v[T::wave()] = best; sh.red_i[T::wave()] = bi; }
            T::sync();
            if (tid == 0) {
                double b = PW_INF;
                int i0 = -1;
                for (int t = 0; t < T::NWAVES; ++t) {
                    int it = sh.red_i[t];
                    if (it == 0x7fffffff) continue;
                    double bt = sh.red_v[t];
                    if (i0 < 0 || bt < b || (bt == b && it < i0)) { b = bt; i0 = it; }
                }
                sh.start = i0;
            }
            T::sync();
        }
        RB_TICK(3);
        if (sh.start < 0) break;
        // ---- walks that can be predicted -----------------------------------------------------
        // A molecule wrapped through the cell faces is walked once from every fragment: each walk
        // finds the same molecule shifted by a lattice vector, and all but one are dropped by the
        // centre-of-mass test.  After the first (clean) walk the others are known in advance: the
        // shift is the image in which the first walk met the new start atom, the walk would retire
        // exactly the atoms it met in that image, and its centre of mass is the first one minus the
        // shift.  A walk predicted to be dropped - with 1e-6 to spare, against 1e-8 of coordinate
        // rounding - is not made.
        if (fr.rebuild) {
            if (tid == 0) {
                int s0 = sh.start, c = WS.cage_of[s0], skip = 0;
                if (c != 0 && WS.cage_ok[c]) {
                    int o = WS.cage_off[s0], rng = WS.cage_rng[c];
                    int os[3] = {o / 9 - 1, (o / 3) % 3 - 1, o % 3 - 1};
                    bool in_range = true, inside = true, sure = true;
                    for (int a = 0; a < 3; ++a) {
                        int lo = ((rng >> (4 * a)) & 3) - 1, hi = ((rng >> (4 * a + 2)) & 3) - 1;
                        if (lo - os[a] < -1 || hi - os[a] > 1) in_range = false;
                        double f = WS.cage_f[3 * (size_t)c + a] - (double)os[a];
                        if (pw_abs(f - sh.bound[0]) < 1e-6 || pw_abs(f - sh.bound[1]) < 1e-6) sure = false;
                        if (!(f >= sh.bound[0] && f < sh.bound[1])) inside = false;
                    }
                    if (in_range && sure && !inside) { skip = 1; sh.skip_cage = c; sh.skip_off = o; }
                }
                sh.skip = skip;
            }
            T::sync();
            if (sh.skip) {
                const int c = sh.skip_cage, o = sh.skip_off;
                for (int q = tid; q < n; q += T::SIZE)
                    if (WS.cage_of[q] == c && WS.cage_off[q] == o) WS.remaining[q] = 0;
                T::sync();
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
                ++rb_skipped;
#endif
                continue;
            }
        }
        ++mol_serial;
        // (read before anybody records anything: the molecule's atoms are entered at its end)
        const bool first_walk = fr.rebuild && WS.cage_of[sh.start] == 0;
        if (tid == 0) {
            sh.lwork[0][0] = sh.start; sh.n_work = 1; sh.n_final = 0; sh.n_next = 0;
            sh.wflags = 0;
            for (int a = 0; a < 3; ++a) { sh.off_lo[a] = 1; sh.off_hi[a] = -1; }
        }
        if (use_bits)
            for (int i = tid; i < WS.bit_words; i += T::SIZE) bits_final[i] = 0;
        T::sync();
        // breadth-first layers (utilities.py:982-1055).  Two team barriers per chunk of a layer:
        //   A  the layer's atoms join the molecule; one wave per atom finds and orders its bonded
        //      neighbours (rb_expand + sort);
        //   C  thread 0 merges the hits in list order (unique, then "not in the molecule") into the
        //      next layer while the other threads retire the layer's atoms from the atom list.
        int cur = 0;                       // which team-shared copy holds the current layer
#if defined(__HIP_DEVICE_COMPILE__)
        // on the device, with the visit bit sets in team-shared memory: the whole walk by one wave, no barriers
        const bool wave_walk = T::WSIZE == 64 && use_bits;
        if (wave_walk) {
            if (T::wave() == 0) {
#if defined(PW_RB_PROFILE)
                long long* const wp = rb_xp;
#else
                long long* const wp = nullptr;
#endif
                const int nl = lists_fast ? rb_wave_walk<T>(fr, WS, (const PW_LDS int*)nb_off, (const PW_LDS int*)nb_ent, sh.start, wp)
                                          : rb_wave_walk<T>(fr, WS, nb_off, nb_ent, sh.start, wp);
                (void)nl;
#if defined(PW_RB_PROFILE)
                rb_layers += nl;
#endif
            }
            T::sync();
            RB_TICK(5);
        }
#else
        const bool wave_walk = false;
#endif
        for (; !wave_walk;) {
            const int nw = sh.n_work;
            if (nw == 0) break;
            ++layer_serial;
            const int nf = sh.n_final;
            const PW_LDS int* wl = sh.lwork[cur];
            PW_LDS int* wl_next = sh.lwork[cur ^ 1];
            for (int k = tid; k < nw; k += T::SIZE) {
                int id = k < RB_LWORK ? wl[k] : work[k];
                if (nf + k < RB_LFINAL) sh.lfinal[nf + k] = id;
                else WS.final_[nf + k] = id;
                if (use_bits) rb_atomic_or64((unsigned long long*)&bits_final[id >> 6], 1ull << (id & 63));
                else WS.stamp_final[id] = mol_serial;
            }
            if (use_bits)
                for (int i = tid; i < WS.bit_words; i += T::SIZE) bits_temp[i] = 0;
            for (int c0 = 0; c0 < nw; c0 += RB_CHUNK) {
                const int cn = nw - c0 < RB_CHUNK ? nw - c0 : RB_CHUNK;
                const bool last = c0 + RB_CHUNK >= nw;
                for (int k = T::wave(); k < cn; k += T::NWAVES) {
                    if (T::lane() == 0) seg_cnt[k] = 0;
                    T::wave_sync();
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
                    rb_expand<T>(fr, WS, nb_off, nb_ent, &w.status, (int*)&sh.wflags, c0 + k < RB_LWORK ? wl[c0 + k] : work[c0 + k], k, T::wave() == 0 ? rb_xp : nullptr);
#else
                    rb_expand<T>(fr, WS, nb_off, nb_ent, &w.status, (int*)&sh.wflags, c0 + k < RB_LWORK ? wl[c0 + k] : work[c0 + k], k);
#endif
                    T::wave_sync();
                    if (T::lane() == 0) {
                        // the atom's hits in list-position order
                        int m = seg_cnt[k] < RB_SEG_CAP ? seg_cnt[k] : RB_SEG_CAP;
                        PW_LDS long long* sgm = &seg[(size_t)k * RB_SEG_CAP];
                        for (int a2 = 1; a2 < m; ++a2) {
                            long long v = sgm[a2];
                            int b2 = a2 - 1;
                            while (b2 >= 0 && sgm[b2] > v) { sgm[b2 + 1] = sgm[b2]; --b2; }
                            sgm[b2 + 1] = v;
                        }
                    }
                }
                T::sync();
                RB_TICK(5);
                // atom_list.remove(i) for the atoms of this layer (utilities.py:1037-1039), after
                // the last of them has been expanded
                if (last)
                    for (int k = tid; k < nw; k += T::SIZE) {
                        int id = k < RB_LWORK ? wl[k] : work[k];
                        if (id < n) WS.remaining[id] = 0;
                    }
                // unique(working_list_temp), then "not in final_molecule" (utilities.py:1044-1055): the
                // hits of the chunk in (atom, position) order; the first occurrence of an id survives
                bool merged = false;
#if defined(__HIP_DEVICE_COMPILE__)
                if (T::WSIZE == 64 && use_bits && T::wave() == 0) {
                    // wave 0, one hit per lane (chunks with up to 64 hits; anything larger goes the serial way)
                    const int lane = T::lane();
                    int mk = 0;
                    if (lane < cn) { int c = seg_cnt[lane]; mk = c < RB_SEG_CAP ? c : RB_SEG_CAP; }
                    int incl = mk;                                   // inclusive prefix of the hit counts
                    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
                    const int H = __shfl(incl, 63);
                    if (H <= 64) {
                        int k = 0;                                   // the atom lane's hit belongs to
                        for (int j = 0; j < cn; ++j) k += (__shfl(incl, j) <= lane) ? 1 : 0;
                        const bool act = lane < H;
                        const int first = __shfl(incl - mk, act ? k : 0);
                        int id = -1 - lane;                          // (distinct dummies for idle lanes)
                        if (act) id = (int)(seg[(size_t)k * RB_SEG_CAP + (lane - first)] & 0xffffffffll);
                        bool dup = false;
                        for (int j = 0; j < H; ++j) { int idj = __shfl(id, j); dup = dup || (j < lane && idj == id); }
                        bool fresh = false, keep = false;
                        if (act && !dup) {
                            unsigned long long bit = 1ull << (id & 63);
                            fresh = !(bits_temp[id >> 6] & bit);
                            keep = fresh && !(bits_final[id >> 6] & bit);
                            if (fresh) rb_atomic_or64((unsigned long long*)&bits_temp[id >> 6], bit);
                        }
                        const unsigned long long bal = T::ballot(keep);
                        const int nn0 = sh.n_next;
                        const int pos = nn0 + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                        if (keep) {
                            if (pos < RB_LWORK) wl_next[pos] = id;
                            else work_next[pos] = id;
                        }
                        if (lane == 0) {
                            const int nn = nn0 + __builtin_popcountll(bal);
                            if (last) { sh.n_final = nf + nw; sh.n_work = nn; sh.n_next = 0; }
                            else sh.n_next = nn;
                        }
                        merged = true;
                    }
                }
#endif
                if (tid == 0 && !merged) {
                    int nn = sh.n_next;
                    for (int k = 0; k < cn; ++k) {
                        int m = seg_cnt[k] < RB_SEG_CAP ? seg_cnt[k] : RB_SEG_CAP;
                        const PW_LDS long long* sgm = &seg[(size_t)k * RB_SEG_CAP];
                        for (int a2 = 0; a2 < m; ++a2) {
                            int id = (int)(sgm[a2] & 0xffffffffll);
                            if (use_bits) {
                                unsigned long long bit = 1ull << (id & 63);
                                unsigned long long seen = bits_temp[id >> 6];
                                if (seen & bit) continue;
                                bits_temp[id >> 6] = seen | bit;
                                if (bits_final[id >> 6] & bit) continue;
                            } else {
                                if (WS.stamp_temp[id] == layer_serial) continue;
                                WS.stamp_temp[id] = layer_serial;
                                if (WS.stamp_final[id] == mol_serial) continue;
                            }
                            if (nn < RB_LWORK) wl_next[nn] = id;
                            else work_next[nn] = id;
                            ++nn;
                        }
                    }
                    if (last) {
                        sh.n_final = nf + nw;
                        sh.n_work = nn;
                        sh.n_next = 0;
                    } else {
                        sh.n_next = nn;
                    }
                }
                T::sync();
                RB_TICK(7);
            }
            { int* t = work; work = work_next; work_next = t; }     // every thread, its own copies
            cur ^= 1;
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            ++rb_layers;
#endif
        }
        // ---- close the molecule ------------------------------------------------------------
        const int m = sh.n_final;
        bool keep = true;
        if (fr.rebuild) {
            // centre of mass of the molecule in fractional coordinates, rounded to 8 places
            // (np.around: x * 1e8 -> rint -> / 1e8), inside [bound0, bound1) on all three axes
            if (m <= WS.term_cap) {
                // terms staged in team-shared memory by all threads, then summed in list order
                PW_LDS double* tx = terms;
                const int tc = WS.term_cap;
                for (int k = tid; k < m; k += T::SIZE) {
                    int q, ax, ay, az;
                    const double* P;
                    rb_decode(WS, n, (k < RB_LFINAL ? sh.lfinal[k] : WS.final_[k]), &q, &ax, &ay, &az, &P);
                    double mq = fr.mass[q];
                    tx[k] = P[0] * mq; tx[tc + k] = P[1] * mq; tx[2 * tc + k] = P[2] * mq; tx[3 * tc + k] = mq;
                }
                T::sync();
                for (int col = tid; col < 3; col += T::SIZE)
                    sh.com[col] = seq_sum_blocked(m, [&](int k) { return tx[col * tc + k]; });
                if (tid == SUM_THREAD) sh.red_v[0] = np_sum_lean((const double*)(tx + 3 * tc), m);
            } else {
                for (int k = tid; k < m; k += T::SIZE) {
                    int q, ax, ay, az;
                    const double* P;
                    rb_decode(WS, n, (k < RB_LFINAL ? sh.lfinal[k] : WS.final_[k]), &q, &ax, &ay, &az, &P);
                    WS.msum[k] = fr.mass[q];
                }
                T::sync();
                for (int col = tid; col < 3; col += T::SIZE)
                    sh.com[col] = seq_sum_blocked(m, [&](int k) {
                        int q, ax, ay, az;
                        const double* P;
                        rb_decode(WS, n, (k < RB_LFINAL ? sh.lfinal[k] : WS.final_[k]), &q, &ax, &ay, &az, &P);
                        return P[col] * fr.mass[q];
                    });
                if (tid == SUM_THREAD) sh.red_v[0] = np_sum_lean(WS.msum, m);
            }
            T::sync();
            if (tid == 0) {
                double total = sh.red_v[0];
                double cf[3];
                rb_mat3(fr.lattice_inv, sh.com[0] / total, sh.com[1] / total, sh.com[2] / total, cf);
                bool in = true;
                for (int c = 0; c < 3; ++c) {
                    double r = __builtin_rint(cf[c] * 1e8) / 1e8;
                    in = in && (r >= sh.bound[0]) && (r < sh.bound[1]);
                    sh.cf[c] = cf[c];
                }
                sh.red_i[0] = in ? 1 : 0;
            }
            T::sync();
            keep = sh.red_i[0] != 0;
            T::sync();
            // first walk through this molecule: remember where it met every atom
            if (first_walk) {
                for (int k = tid; k < m; k += T::SIZE) {
                    int q, ax, ay, az;
                    const double* P;
                    rb_decode(WS, n, (k < RB_LFINAL ? sh.lfinal[k] : WS.final_[k]), &q, &ax, &ay, &az, &P);
#if defined(__HIP_DEVICE_COMPILE__)
                    int old = atomicExch(&WS.cage_of[q], mol_serial);
#else
                    int old = WS.cage_of[q];
                    WS.cage_of[q] = mol_serial;
#endif
                    if (old != 0) rb_atomic_or((int*)&sh.wflags, RB_WF_REPEAT);
                    WS.cage_off[q] = (unsigned char)((ax + 1) * 9 + (ay + 1) * 3 + (az + 1));
                    int ao[3] = {ax, ay, az};
                    for (int a = 0; a < 3; ++a) {
#if defined(__HIP_DEVICE_COMPILE__)
                        atomicMin((int*)&sh.off_lo[a], ao[a]);
                        atomicMax((int*)&sh.off_hi[a], ao[a]);
#else
                        if (ao[a] < sh.off_lo[a]) sh.off_lo[a] = ao[a];
                        if (ao[a] > sh.off_hi[a]) sh.off_hi[a] = ao[a];
#endif
                    }
                }
                T::sync();
                if (tid == 0) {
                    int rng = 0;
                    for (int a = 0; a < 3; ++a) {
                        rng |= (sh.off_lo[a] + 1) << (4 * a);
                        rng |= (sh.off_hi[a] + 1) << (4 * a + 2);
                        WS.cage_f[3 * (size_t)mol_serial + a] = sh.cf[a];
                    }
                    WS.cage_rng[mol_serial] = rng;
                    WS.cage_ok[mol_serial] = sh.wflags == 0 ? 1 : 0;
                }
                T::sync();
            }
        }
        RB_TICK(9);
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        ++rb_mols;
#endif
        if (keep) {
            const int base = w.n_out;
            const bool fits = base + m <= out.atoms_cap && w.n_mol < out.mols_cap;
            if (fits) {
                for (int k = tid; k < m; k += T::SIZE) {
                    int q, ax, ay, az;
                    const double* P;
                    int id = (k < RB_LFINAL ? sh.lfinal[k] : WS.final_[k]);
                    rb_decode(WS, n, id, &q, &ax, &ay, &az, &P);
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
    RB_TICK(10);
    if (tid == 0) {
        *out.n_mol = w.n_mol;
        *out.status = w.status;
    }
    T::sync();
#if defined(PW_RB_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (tid == 0 && blockIdx.x == 0)
        printf("RBPROF us: setup %lld pre %lld cand %lld start %lld append %lld expand %lld sort %lld unique %lld "
               "layer_end %lld close %lld emit %lld | layers %d walks %d kept %d\n",
               rb_prof[0] / 100, rb_prof[1] / 100, rb_prof[2] / 100, rb_prof[3] / 100, rb_prof[4] / 100, rb_prof[5] / 100,
               rb_prof[6] / 100, rb_prof[7] / 100, rb_prof[8] / 100, rb_prof[9] / 100, rb_prof[10] / 100, rb_layers, rb_mols,
               w.n_mol);
    if (tid == 0 && blockIdx.x == 0) printf("RBPROF wave 0 candidate scan: %d images, %d rounds; walks predicted and skipped: %d\n", rb_imgs, rb_rounds, rb_skipped);
    if (tid == 0 && blockIdx.x == 0)
        printf("RBPROF walk (one wave) us: join+clear %lld, layer atoms+lists %lld, scan+entry atoms %lld, entries %lld, hits+merge %lld, retire %lld\n",
               rb_xp[0] / 100, rb_xp[1] / 100, rb_xp[2] / 100, rb_xp[3] / 100, rb_xp[4] / 100, rb_xp[5] / 100);
#endif
}

}  // namespace pw
