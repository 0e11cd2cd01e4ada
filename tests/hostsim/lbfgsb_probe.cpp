// tests/hostsim/lbfgsb_probe.cpp -- exposes pw::Lbfgsb<N> step by step so the
// Python test can run SciPy's own _lbfgsb.setulb in lockstep and compare every
// intermediate.  Test infrastructure only.
#include "../../pywindow_amd/csrc/pw_lbfgsb.hpp"
#include <new>
#include <string.h>
using namespace pw;
template <int N> static void dump(Lbfgsb<N>* s, double* wa_out, int* ints, double* dbl) {
    const int m = LB_M;
    double* p = wa_out;
    memcpy(p, s->ws, sizeof(double) * m * N); p += m * N;
    memcpy(p, s->wy, sizeof(double) * m * N); p += m * N;
    memcpy(p, s->sy, sizeof(double) * m * m); p += m * m;
    memcpy(p, s->ss, sizeof(double) * m * m); p += m * m;
    memcpy(p, s->wt, sizeof(double) * m * m); p += m * m;
    memcpy(p, s->wn, sizeof(double) * 4 * m * m); p += 4 * m * m;
    memset(p, 0, sizeof(double) * 4 * m * m); p += 4 * m * m;   // WN1 shares wn's lower triangle (not compared)
    memcpy(p, s->z, sizeof(double) * N); p += N;
    memcpy(p, s->r, sizeof(double) * N); p += N;
    memcpy(p, s->d, sizeof(double) * N); p += N;
    memcpy(p, s->t, sizeof(double) * N); p += N;
    memcpy(p, s->xp, sizeof(double) * N); p += N;
    memcpy(p, s->wa, sizeof(double) * 8 * m); p += 8 * m;
    ints[0] = s->task; ints[1] = s->msg; ints[2] = s->col; ints[3] = s->head; ints[4] = s->iter;
    ints[5] = s->iupdat; ints[6] = s->nfgv; ints[7] = s->ifun; ints[8] = s->iback; ints[9] = s->info;
    ints[10] = s->nfree; ints[11] = s->nskip; ints[12] = s->updatd;
    dbl[0] = s->theta; dbl[1] = s->stp; dbl[2] = s->gd; dbl[3] = s->gdold; dbl[4] = s->dtd;
    dbl[5] = s->dnorm; dbl[6] = s->sbgnrm; dbl[7] = s->stpmx; dbl[8] = s->fold;
}
#define DEF(N)                                                                               \
    extern "C" void* hs_lb##N##_new(const double* x0, const double* l, const double* u,      \
                                    const int* nbd, double factr, double pgtol, int maxls) { \
        auto* s = new Lbfgsb<N>();                                                           \
        memset((void*)s, 0, sizeof(*s));                                                     \
        auto* m = new LbMem<N>();                                                            \
        memset((void*)m, 0xff, sizeof(*m));   /* setup() must not rely on what the block held */  \
        s->template setup<HostTeam>(m, x0, l, u, nbd, factr, pgtol, maxls);                     \
        return s;                                                                            \
    }                                                                                        \
    extern "C" void hs_lb##N##_free(void* h) { delete ((Lbfgsb<N>*)h)->mem; delete (Lbfgsb<N>*)h; }                       \
    extern "C" void hs_lb##N##_step(void* h, double* x, double f, const double* g,          \
                                    int set_fg) {                                            \
        auto* s = (Lbfgsb<N>*)h;                                                             \
        if (set_fg) { s->f = f; for (int i = 0; i < N; ++i) s->g[i] = g[i]; }                \
        s->template step<HostTeam>();                                                                         \
        for (int i = 0; i < N; ++i) x[i] = s->x[i];                                          \
    }                                                                                        \
    extern "C" void hs_lb##N##_dump(void* h, double* wa, int* ints, double* dbl) {           \
        dump<N>((Lbfgsb<N>*)h, wa, ints, dbl);                                               \
    }
// ---- the two drivers on ONE objective (round 5's advisor: the product runs minimize(), the lockstep tests drive
// step(): nothing compared them).  cb(x, &f, g, user) is the caller's objective and gradient; SciPy's ScalarFunction
// answers a repeated x from its cache, and so does this wrapper -- for both drivers alike.  `per_call`: what one
// evaluation adds to nfev (4 for the forward-difference pore objective: f and three difference points).
typedef void (*hs_fg_cb)(const double* x, double* f, double* g, void* user);
template <int N> struct CbObjective {
    hs_fg_cb cb; void* user; int per_call; int nfev; bool have; double lx[N], lf, lg[N];
    void operator()(const double* xq, double& fo, double* go) {
        bool same = have;
        for (int i = 0; i < N && same; ++i) same = xq[i] == lx[i];
        if (!same) {
            cb(xq, &lf, lg, user);
            for (int i = 0; i < N; ++i) lx[i] = xq[i];
            have = true;
            nfev += per_call;
        }
        fo = lf;
        for (int i = 0; i < N; ++i) go[i] = lg[i];
    }
};
// mode 0: Lbfgsb::minimize (what the kernels and the host path run); mode 1: SciPy's loop around step()
// (_lbfgsb_py.py:427-456: f and g supplied on FG, the limits tested at a new iterate only).  out_int: nit, nfev,
// task, msg; then the state dump of hs_lbN_dump.
template <int N> static void drive(int mode, const double* x0, const double* l, const double* u, const int* nbd, double factr,
                                   double pgtol, int maxls, hs_fg_cb cb, void* user, int per_call, int maxiter, int maxfun,
                                   double* x_out, double* f_out, int* out_int, double* wa, int* ints, double* dbl) {
    auto* s = new Lbfgsb<N>();
    memset((void*)s, 0, sizeof(*s));
    auto* m = new LbMem<N>();
    memset((void*)m, 0xff, sizeof(*m));
    s->template setup<HostTeam>(m, x0, l, u, nbd, factr, pgtol, maxls);
    CbObjective<N> fg{cb, user, per_call, 0, false, {}, 0.0, {}};
    int nit = 0;
    if (mode == 0) {
        s->template minimize<HostTeam>(fg, maxiter, maxfun, &nit);
    } else {
        for (;;) {
            s->template step<HostTeam>();
            if (s->task == LB_FG) {
                double xv[N], gv[N], fv;
                for (int i = 0; i < N; ++i) xv[i] = s->x[i];
                fg(xv, fv, gv);
                s->f = fv;
                for (int i = 0; i < N; ++i) s->g[i] = gv[i];
            } else if (s->task == LB_NEW_X) {
                nit += 1;
                if (nit >= maxiter || fg.nfev > maxfun) break;
            } else {
                break;
            }
        }
    }
    for (int i = 0; i < N; ++i) x_out[i] = s->x[i];
    *f_out = s->f;
    out_int[0] = nit; out_int[1] = fg.nfev; out_int[2] = s->task; out_int[3] = s->msg;
    dump<N>(s, wa, ints, dbl);
    delete s->mem;
    delete s;
}
#define DEFDRIVE(N)                                                                                                       \
    extern "C" void hs_lb##N##_drive(int mode, const double* x0, const double* l, const double* u, const int* nbd,     \
                                     double factr, double pgtol, int maxls, hs_fg_cb cb, void* user, int per_call,      \
                                     int maxiter, int maxfun, double* x_out, double* f_out, int* out_int, double* wa,   \
                                     int* ints, double* dbl) {                                                          \
        drive<N>(mode, x0, l, u, nbd, factr, pgtol, maxls, cb, user, per_call, maxiter, maxfun, x_out, f_out, out_int,  \
                 wa, ints, dbl);                                                                                        \
    }
DEF(1)
DEF(2)
DEF(3)
DEFDRIVE(1)
DEFDRIVE(2)
DEFDRIVE(3)
