/*
 * pywindow_amd.h -- C ABI of libpywindow_hip.so, the MI355X (gfx950) engine for
 * pywindow's per-molecule structural analysis.
 *
 * The reference (marcinmiklitz/pywindow) is pure Python and has no FFI layer; the
 * boundary this library stands behind is the set of free functions that
 * src/pywindow/_internal/molecular.py:29-44 imports from utilities.py and that
 * Molecule.full_analysis() chains (molecular.py:156-202).  One *unit* is one
 * (frame, molecule); a batch is every unit of a trajectory, analysed by one
 * kernel launch per GPU.  Each entry point below names the reference function
 * it replaces.
 *
 * Conventions: plain pointers and sizes, caller-allocated buffers, no
 * exceptions across the ABI -- every function returns 0 on success or a
 * negative PW_E_* code.  All arithmetic is IEEE double.  There is no CPU
 * FALLBACK: without a usable HIP device every compute call on a device context
 * fails with PW_E_NO_DEVICE.  (A host context, pw_context_create(-1, ..), is an
 * explicit choice of the caller and runs the same source on host threads.)
 *
 * Threads (the reference's workers are stateless processes,
 * trajectory.py:564-582).  Every entry point that takes a pw_context holds that
 * context's mutex for the duration of the call: any number of threads may call
 * into ONE context, their calls run one after the other, and different contexts
 * share no mutable state -- two threads with a context each (even on the same
 * device) never wait for one another in the library.  pw_last_error() is per
 * thread.  What the library cannot see is which calls belong together.  Three
 * pieces of per-context state outlive a call, and a caller that shares a context
 * between threads keeps each sequence together itself (the Python binding holds
 * Context.lock across them):
 *   - the page-locked staging buffer: pw_context_pinned .. pw_resident_upload
 *     (the buffer is free again when the upload returns);
 *   - "the records fetched last": pw_resident_download / pw_resident_extra_windows
 *     .. pw_context_extra_windows;
 *   - the knobs: pw_context_set_params .. the launches that should see them.
 * pw_history handles are read-only after pw_history_open and may be read from
 * any number of threads; the reader decodes on a team of host threads started
 * once per process, and pw_history_stream_read runs one more thread of its own
 * for the duration of the call (the appends, which take the context's mutex one
 * at a time like any other caller).  A second live device context on a device runs its
 * analyses as single launches (pw_context_pipelined), not as the pipeline.
 */
#ifndef PYWINDOW_AMD_H
#define PYWINDOW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PW_W_MAX 16       /* windows held by the fixed-size record; a molecule with more keeps the first
                             PW_W_MAX there, sets PW_ST_WINDOW_OVERFLOW and hands the rest over through
                             pw_context_extra_windows -- the reference has no limit (utilities.py:1526-1536) */
#define PW_DBSCAN_MAX 8192 /* points pw_dbscan accepts */
#define PW_P_MAX 2048     /* sampling vectors the team workspaces are sized for AT LEAST; the capacity follows
                             the `adjust` knobs of pw_params (the reference's count is unbounded,
                             utilities.py:1409, 1616) and pw_analysis_batch grows it further when a unit asks */

/* error codes */
#define PW_OK 0
#define PW_E_NO_DEVICE (-1)
#define PW_E_BAD_ARG (-2)
#define PW_E_HIP (-3)
#define PW_E_TOO_LARGE (-4) /* a molecule has more atoms than fit in LDS */
#define PW_E_NOMEM (-5)
#define PW_E_RETRY (-6)     /* a capacity was grown for this batch: launch the analysis again, then download */
#define PW_E_TIMEOUT (-7)   /* a launch of the pipeline gave up waiting for another one (pw_last_error names the wait);
                               the records are incomplete: repeat the analysis (pw_context_retries counts repeats).
                               Every wait inside a kernel is bounded by time WITHOUT PROGRESS of the other side:
                               PW_WAIT_LIMIT_MS (default 250) between launches of one analysis -- a window team that sees
                               no unit published, a residency gate that sees no optimiser team start -- and
                               PW_STREAM_LIMIT_MS (default 5000) for the host's next append to a streamed batch; both are
                               read by pw_context_create.  A wait during which the other side keeps making progress (a
                               device shared with another tenant, a batch of long chains) is never cut short. */

/* stage selection bits for pw_analysis_* */
#define PW_STAGE_BASIC 1u   /* molecular_weight, center_of_mass, max_dim, pore_diameter */
#define PW_STAGE_AVG 2u     /* find_average_diameter */
#define PW_STAGE_OPT 4u     /* opt_pore_diameter */
#define PW_STAGE_WINDOWS 8u /* find_windows (implies PW_STAGE_OPT) */
#define PW_STAGE_ALL 15u

/* per-unit status bits (pw_unit_out.status) */
#define PW_ST_OK 0
#define PW_ST_NEGATIVE_PORE 1      /* pore radius <= 0: the reference's bounds would be inverted */
#define PW_ST_WINDOW_OVERFLOW 2    /* more than PW_W_MAX windows: n_windows is the true count, the record holds the
                                      first PW_W_MAX, the others are in the context's extra-window list */
#define PW_ST_POINTS_OVERFLOW 4    /* more sampling vectors than the workspace of this launch holds (n_points /
                                      n_points_avg say how many): avg_d is NaN / n_windows -1, NOT a result */
#define PW_ST_WINDOW_DROPPED 8     /* a cluster's refined path scan failed (reference: None + warning) */
#define PW_ST_WINDOW_NEGATIVE 16   /* a window diameter < 0 (reference: warning) */
#define PW_ST_Z_BOUNDS 32          /* z_bounds upper < -new_z with lb_z (reference: scipy raises ValueError) */
#define PW_ST_TOO_FEW_POINTS 64    /* fewer than 10 sampling vectors in find_windows (reference: KDTree.query(k=10)
                                      raises ValueError, utilities.py:1428-1431) */
#define PW_ST_PATH_TOO_LONG 128    /* find_windows: a sampling vector's path would have more than 2^20 points (sphere radius /
                                      increment, or / increment2): a pore centre that an open or enormous search box let run
                                      away.  The reference builds such a path as a Python list (MemoryError, or hours);
                                      no windows are computed for the unit */

/* Input batch: ragged molecules, atoms of unit u are [atom_offset[u], atom_offset[u+1]). */
typedef struct pw_batch_in {
    int64_t n_units;
    const int64_t *atom_offset; /* n_units + 1 */
    const double *xyz;          /* sum(N) x 3, row-major as numpy (N,3) */
    const double *vdw;          /* per atom: atomic_vdw_radius[element] (tables.py:111-197) */
    const double *mass;         /* per atom: atomic_mass[element]      (tables.py:22-108)  */
    int64_t template_atoms;     /* 0: vdw / mass have one entry per atom of the batch (sum(N));
                                   T > 0: every unit has T atoms and vdw / mass are ONE template of T
                                   entries -- the frames of a trajectory share their elements
                                   (trajectory.py:245-248), so the constants travel once */
} pw_batch_in;

/* Fixed-size result record of one unit == Molecule.properties (molecular.py:215-352). */
typedef struct pw_unit_out {
    int32_t n_atoms;
    int32_t status;
    double mw;             /* molecular_weight()        utilities.py:96  */
    double com[3];         /* center_of_mass()          utilities.py:127 */
    double maxd;           /* max_dim()                 utilities.py:355 */
    int32_t maxd_i, maxd_j;
    double avg_d;          /* find_average_diameter()   utilities.py:1586 */
    double pore_d;         /* pore_diameter()           utilities.py:375 */
    int32_t pore_atom;
    int32_t pore_opt_atom;
    double pore_vol;       /* sphere_volume(pore_d/2)   utilities.py:429 */
    double pore_opt_d;     /* opt_pore_diameter()       utilities.py:400 */
    double pore_opt_c[3];
    double pore_vol_opt;
    int32_t n_windows;     /* find_windows(): -1 => None, else number of windows  utilities.py:1364 */
    int32_t n_clusters;
    double win_d[PW_W_MAX];
    double win_c[PW_W_MAX][3];
    /* diagnostics (not part of the reference's dict; used by the parity tests) */
    int32_t n_points;      /* sampling vectors in find_windows */
    int32_t n_points_avg;  /* sampling vectors in find_average_diameter */
    int32_t n_survivors;   /* vectors that reach the outside */
    int32_t opt_nit, opt_nfev, opt_task, opt_msg;
    int32_t n_eval;        /* point-vs-molecule evaluations performed */
    double eps;            /* DBSCAN radius */
    double sphere_r;       /* sampling sphere radius in find_windows */
} pw_unit_out;

/* A window beyond the PW_W_MAX the record holds (PW_ST_WINDOW_OVERFLOW): position `index` >= PW_W_MAX in
 * the reference's output order of unit `unit` (utilities.py:1526-1536). */
typedef struct pw_extra_window {
    int64_t unit;
    int32_t index;
    int32_t reserved;
    double d;
    double c[3];
} pw_extra_window;

/* Stage-level capture of find_windows for one unit (pw_analysis_debug; the parity tests compare it
 * with the reference's intermediate results). */
typedef struct pw_unit_debug {
    int32_t n_survivors;            /* sampling vectors that pass vector_analysis (utilities.py:1457-1467) */
    int32_t n_clusters;
    int32_t pass_idx[PW_P_MAX];     /* their indices on the sampling sphere, ascending (the first PW_P_MAX) */
    int32_t labels[PW_P_MAX];       /* DBSCAN label of each (utilities.py:1478-1487) */
    double gap2[PW_P_MAX];          /* vector_analysis result [1]: 2 * narrowest gap along the path */
    double win[PW_W_MAX][12];       /* window_analysis per cluster (utilities.py:1191-1361): chosen vector (3),
                                       angle_1, angle_2 (as angle_between_vectors returns them), new_z,
                                       diameter at the neck point, z optimum,
                                       x, y optimum, final diameter, evaluations */
} pw_unit_debug;

/* Optional knobs of the reference's free functions.  Defaults are the
 * values Molecule.full_analysis() uses; a context starts with the defaults. */
typedef struct pw_params {
    double adjust_windows;  /* find_windows(adjust=1):           sampling density, utilities.py:1410 */
    double adjust_average;  /* find_average_diameter(adjust=1):  sampling density, utilities.py:1615 */
    double increment;       /* find_windows(increment=1.0):      coarse path-scan step, utilities.py:1457 */
    int32_t pore_opt;       /* find_windows(pore_opt=True): centre on the optimised pore, :1380-1393 */
    int32_t opt_flags;      /* opt_pore_diameter(bounds=, com=), utilities.py:400-426: PW_OPT_* bits */
    double opt_x0[3];       /* com=: start of the optimisation (default: the centre of mass) */
    double opt_lo[3];       /* bounds=: lower / upper bound per axis, -/+HUGE_VAL for None */
    double opt_hi[3];       /*          (default: start -/+ the pore radius at the start) */
    /* window_analysis(increment2=0.1, z_bounds=None, lb_z=True, z_second_mini=False), utilities.py:1191-1200 */
    double increment2;      /* refined path-scan step along the chosen vector, :1221-1224 */
    double z_lo, z_hi;      /* z_bounds: -/+HUGE_VAL for None; z_lo is replaced by -new_z when lb_z, :1296-1297 */
    int32_t lb_z;           /* lower bound of the neck search = -new_z (default 1) */
    int32_t z_second_mini;  /* second neck search after the in-plane optimisation, :1326-1334 (default 0) */
} pw_params;
#define PW_OPT_CUSTOM_START 1
#define PW_OPT_CUSTOM_BOUNDS 2

typedef struct pw_context pw_context;   /* device, stream, workspace */
typedef struct pw_resident pw_resident; /* a batch resident in HBM */

int pw_device_count(void);
const char *pw_version(void);
const char *pw_last_error(void);

/* device >= 0: a HIP device ordinal.  device == -1: the explicit HOST path -- the same unit pipeline
 * (pywindow_amd/csrc/pw_unit.hpp, single source with the kernels) compiled for the host and run by threads
 * over the units; it serves pw_analysis_batch / pw_analysis_debug / pw_point_gaps and the pw_resident_*
 * calls (batches then live in host memory), nothing else, and it is never chosen implicitly: a context
 * for a device that does not exist fails with PW_E_NO_DEVICE.  (BASELINE.json configs[0].) */
int pw_context_create(int device, pw_context **ctx);
/* Page-locked host staging buffer owned by the context (at least `bytes`, valid until a later call asks
 * for more): a reader that decodes frames straight into it (pw_history_read) makes the copies of
 * pw_resident_upload asynchronous DMA.  Device contexts only. */
int pw_context_pinned(pw_context *ctx, size_t bytes, void **ptr);
/* threads of a device == -1 context (default: PW_CPU_THREADS or the hardware concurrency); threads <= 0
 * only reports.  Returns 0 for device contexts. */
int pw_context_host_threads(pw_context *ctx, int threads);
void pw_context_destroy(pw_context *ctx);
/* knobs used by every later launch on this context (validated: adjust > 0, increment > 0) */
void pw_params_default(pw_params *params);
int pw_context_set_params(pw_context *ctx, const pw_params *params);

/* Host-buffer path: H2D copy, one launch, D2H copy, synchronous.
 * Replaces the per-frame loop `mol.full_analysis()` of trajectory.py:518-522
 * for `stages == PW_STAGE_ALL`; partial `stages` give the fine-grained calls
 * (pore_diameter, max_dim, ... ) the parity tests exercise one by one. */
int pw_analysis_batch(pw_context *ctx, const pw_batch_in *in, uint32_t stages, pw_unit_out *out);
/* Windows beyond PW_W_MAX of the analysis whose records were fetched last on this context
 * (pw_analysis_batch, pw_analysis_debug, pw_resident_download), ordered by (unit, index): copies at
 * most `cap` entries to buf and returns how many exist.  find_windows has no upper limit on the
 * number of windows (utilities.py:1526-1536); the record keeps the first PW_W_MAX. */
int64_t pw_context_extra_windows(pw_context *ctx, pw_extra_window *buf, int64_t cap);
/* 1 when analyses on this context run as the overlapped two-launch pipeline (optimiser chains | average diameter + window search), 0 when as single launches:
 * the pipeline needs its ten HIP streams to run concurrently, i.e. GPU_MAX_HW_QUEUES >= 10 in the
 * environment BEFORE the process first initialises HIP; pw_context_create measures whether they do and
 * falls back (with a line on stderr) instead of letting gate kernels wait for launches queued behind them */
int pw_context_pipelined(pw_context *ctx);
/* diagnostic: how many gates of the pipeline gave up waiting since the context was created (tail gates |
 * head gates << 16 | residency gates << 32).  Tail and head gates only pace launches: an expiry costs 20 ms and
 * changes no result.  A residency gate that sees no optimiser team start for PW_WAIT_LIMIT_MS lets the launches
 * behind it go when some teams HAVE started (the device is busy; nothing is lost) and gives the analysis up
 * (PW_E_TIMEOUT) when not one has.  Zero on a healthy device. */
int pw_context_gate_timeouts(pw_context *ctx, uint64_t *count);
/* How many analyses were REPEATED on this context after a launch gave up waiting for another one (PW_E_TIMEOUT):
 * by pw_analysis_batch / pw_analysis_debug themselves (up to PW_TIMEOUT_REPEATS times, default 2), and by callers of the pw_resident_*
 * entry points, who repeat on PW_E_TIMEOUT and say so with pw_context_count_retry.  Zero on a healthy device;
 * pw_retries_total() is the same over every context of the process.  (The reference has nothing to compare:
 * its Pool workers either return or raise, trajectory.py:553-586.) */
int pw_context_retries(pw_context *ctx, uint64_t *count);
int pw_context_count_retry(pw_context *ctx);
uint64_t pw_retries_total(void);
/* diagnostic: the hand-off queues of the pipeline's sets as they are now -- out[4 * s + 0..3] = units taken by
 * window teams (head), published by optimiser chains (tail), optimiser teams started, error flag of set s;
 * s < 4 (cap >= 16).  Reads device memory without waiting for anything. */
int pw_context_queue_state(pw_context *ctx, uint64_t *out, int cap);
/* sampling vectors the team workspaces of this context currently hold per molecule (>= PW_P_MAX) */
int pw_context_point_capacity(pw_context *ctx);
/* at least n_points sampling vectors per molecule in the workspaces of every later launch.  pw_analysis_batch
 * does this by itself when a unit carries PW_ST_POINTS_OVERFLOW (the reference's count
 * int(log10(4 pi R^2) * 250 * adjust), utilities.py:1409, 1616, has no upper limit); callers of the
 * pw_resident_* entry points read n_points / n_points_avg of the flagged records, reserve, launch again. */
int pw_context_reserve_points(pw_context *ctx, int64_t n_points);

/* A batch of ONE molecule type whose coordinates arrive in pieces, in unit order, while it is being analysed
 * (the frame loop of the reference reads and analyses frame after frame, trajectory.py:496-522; here the reader
 * feeds a launch that is already running).  begin: sizes known, radii / masses as a template of template_atoms
 * entries; pw_resident_launch may follow at once -- the launch's teams wait for the units they are handed.
 * append: coordinates [count][template_atoms][3] of units first .. first + count - 1, `first` = the number appended
 * so far; the call returns when the copy has landed (from page-locked memory, pw_context_pinned, a DMA of tens of
 * microseconds per megabyte) and the buffer may be reused.  Downloading an incomplete batch is PW_E_BAD_ARG; a
 * launch that sees no unit appended for PW_STREAM_LIMIT_MS (5 s) gives up and the download reports it (PW_E_TIMEOUT).  Device contexts only;
 * molecules beyond LDS (PW_E_TOO_LARGE) go through pw_resident_upload.  On a context that runs an analysis as ONE
 * launch (pw_context_pipelined() == 0), or for stages without the window search, a launch asked for before the last
 * unit has arrived is made by the append that completes the batch (nothing overlaps, nothing waits). */
int pw_resident_stream_begin(pw_context *ctx, int64_t n_units, int64_t template_atoms, const double *vdw,
                             const double *mass, pw_resident **res);
int pw_resident_stream_append(pw_context *ctx, pw_resident *res, const double *xyz, int64_t first, int64_t count);

/* The same analysis with the intermediate results of find_windows captured per unit (dbg: n_units
 * records, caller-allocated).  Test instrumentation: one launch at a time, no overlap. */
int pw_analysis_debug(pw_context *ctx, const pw_batch_in *in, uint32_t stages, pw_unit_out *out,
                      pw_unit_debug *dbg);

/* Fine-grained: min_i(|r_i - p| - vdw_i) and its first argmin at arbitrary points,
 * point q belonging to unit unit_of_point[q]  (pore_diameter(.., com=p)[0]/2 and [1],
 * utilities.py:375-388; the objective of every optimiser on the path). */
int pw_point_gaps(pw_context *ctx, const pw_batch_in *in, const int64_t *unit_of_point,
                  const double *points, int64_t n_points, double *gap, int32_t *argmin);

/* Fine-grained: sklearn.cluster.DBSCAN(eps, min_samples=5).fit(points).labels_ as find_windows calls it
 * (utilities.py:1478-1487; sklearn/cluster/_dbscan_inner.pyx): clusters numbered in the order of their
 * smallest member, border points with the lowest-numbered adjacent cluster, noise -1.  n <= PW_DBSCAN_MAX points
 * (n x 3, row-major).  mode bit 0: a one-wave team instead of four waves; bit 1: every array of the routine
 * in global memory instead of LDS. */
int pw_dbscan(pw_context *ctx, const double *points, int64_t n, double eps, int mode, int32_t *labels,
              int32_t *n_clusters);

/* Fine-grained: numpy's float64 add.reduce over a contiguous array (pairwise blocks of <= 128 with eight
 * accumulators, 8192-element buffers) as one team computes it -- the order behind np.mean / np.sum in
 * utilities.py:1434 (mean of the k-NN distances) and :1650 (mean of the ray exits).  mode bit 0: a
 * one-wave team instead of four waves; bit 1: the team's scratch in global memory instead of LDS. */
int pw_pairwise_sum(pw_context *ctx, const double *values, int64_t n, int mode, double *sum);

/* Resident path (inputs stay in HBM between launches; used by bench.py and the
 * trajectory driver when several analyses run on the same frames). */
int pw_resident_upload(pw_context *ctx, const pw_batch_in *in, pw_resident **res);
int pw_resident_launch(pw_context *ctx, pw_resident *res, uint32_t stages); /* async on ctx stream */
int pw_resident_sync(pw_context *ctx);
int pw_resident_download(pw_context *ctx, pw_resident *res, pw_unit_out *out);
void pw_resident_free(pw_context *ctx, pw_resident *res);
/* `iters` back-to-back launches bracketed by HIP events on the launch stream;
 * writes the average milliseconds per launch. */
int pw_resident_time(pw_context *ctx, pw_resident *res, uint32_t stages, int iters, float *ms_per_launch);
/* one analysis on its own, timed per launch of the pipeline with HIP events on the launch's own
 * stream: ms[0] optimiser chains, ms[1] average diameter (0: since round 6 that stage runs inside the window teams
 * unless PW_B_LAUNCH=1 asks for its own launch), ms[2] window search (measurement only) */
int pw_resident_stage_times(pw_context *ctx, pw_resident *res, float *ms3);
/* raw device pointer of the result records (for RCCL gathers by the host side) */
void *pw_resident_device_results(pw_resident *res);
/* For callers that read the records on the device (pw_resident_results_ready): waits for the latest
 * launch of the batch, fails with PW_E_TIMEOUT if its window launch timed out, and loads its windows beyond
 * PW_W_MAX into the context's list (*count of them; pw_context_extra_windows reads them).  PW_E_RETRY: the
 * device list for them has just been allocated -- launch again.  pw_resident_download does all of this. */
int pw_resident_extra_windows(pw_context *ctx, pw_resident *res, int64_t *count);
/* Stream-ordered hand-over of the latest launch's records to a stream of the caller (hipStream_t as
 * an opaque pointer; NULL = the legacy default stream): work queued on `stream` afterwards sees the
 * finished records at *results (device pointer), without a host synchronisation.  This is how the
 * one collective of a multi-GPU run -- the gather of the records, Trajectory._analysis_parallel's
 * pool.get() (trajectory.py:553-586) -- reads them: no host bounce. */
int pw_resident_results_ready(pw_context *ctx, pw_resident *res, void *stream, void **results);
/* ... and the way back: the launch that next overwrites those records waits for everything queued
 * on `stream` so far (call it after queueing the gather). */
int pw_resident_results_release(pw_context *ctx, pw_resident *res, void *stream);
int64_t pw_resident_units(pw_resident *res);
/* the HIP stream (hipStream_t) launches are issued on, as an opaque pointer */
void *pw_context_stream(pw_context *ctx);

/* ---- periodic pre-processing ---------------------------------------------
 * create_supercell (utilities.py:768-810) + discrete_molecules (utilities.py:820-1085) as
 * driven by MolecularSystem.rebuild_system / make_modular (molecular.py:672-708, 798-824):
 * split every frame of a system into discrete molecules; with `rebuild`, molecules wrapped
 * across the cell faces are re-assembled from the 3x3x3 supercell and the copies whose
 * centre of mass lies outside the cell are dropped.  Atom order inside each molecule and
 * the order of the molecules are the reference's.  One topology (covalent radii, masses,
 * terminal flags) for all frames; coordinates and lattice per frame. */
typedef struct pw_cell_in {
    int64_t n_frames;
    int32_t n_atoms;           /* atoms per frame */
    int32_t rebuild;           /* 1: rebuild through the periodic boundary (needs lattice) */
    const double *xyz;         /* n_frames x n_atoms x 3, as loaded (not rounded) */
    const double *lattice;     /* n_frames x 9 row-major lattice matrices; NULL = non-periodic */
    const double *lattice_inv; /* n_frames x 9: numpy.linalg.inv(lattice), utilities.py:726 */
    const double *cov;         /* n_atoms: atomic_covalent_radius[element] (tables.py:200-286) */
    const double *mass;        /* n_atoms */
    const uint8_t *terminal;   /* n_atoms: 1 if the element ends a bond path (utilities.py:943) */
    double max_dist;           /* 2 * max covalent radius present + tol (utilities.py:949-953) */
    double tol;                /* bond tolerance, 0.4 */
} pw_cell_in;

#define PW_RB_NB_OVERFLOW 1     /* more than 32 atoms within max_dist of one atom */
#define PW_RB_SEG_OVERFLOW 2    /* more than 32 bonded neighbours of one atom */
#define PW_RB_ATOMS_OVERFLOW 4  /* atoms_cap too small (retry with a larger one) */
#define PW_RB_MOLS_OVERFLOW 8   /* mols_cap too small */
#define PW_RB_THIN_CELL 16      /* a cell height is below max_dist */

typedef struct pw_cell_out {   /* caller-allocated */
    int32_t atoms_cap;         /* output atoms per frame */
    int32_t mols_cap;          /* output molecules per frame */
    int32_t *n_mol;            /* n_frames */
    int32_t *status;           /* n_frames: PW_RB_* bits */
    int32_t *mol_offset;       /* n_frames x (mols_cap + 1): atoms of molecule m are [off[m], off[m+1]) */
    int32_t *src_atom;         /* n_frames x atoms_cap: index of the input atom */
    int8_t *src_image;         /* n_frames x atoms_cap: -1 = the input atom itself, else image 0..26
                                  (a, b, c nested, 13 = the cell) */
    double *xyz;               /* n_frames x atoms_cap x 3: coordinates rounded to 8 decimals */
} pw_cell_out;

int pw_discrete_molecules(pw_context *ctx, const pw_cell_in *in, const pw_cell_out *out);
/* The same, but the molecules of all frames stay on the device as ONE resident batch (units in
 * frame order, then molecule order) ready for pw_resident_launch: the modular branch of
 * Trajectory._analysis_serial (trajectory.py:512-522) without a host round trip.  `vdw`: van der
 * Waals radius per input atom.  n_mol / status (n_frames each) are returned to the host; *res is
 * NULL when no frame has a molecule.  PW_E_TOO_LARGE: a frame needs larger caps (retry). */
int pw_resident_from_cells(pw_context *ctx, const pw_cell_in *in, const double *vdw, int32_t atoms_cap,
                           int32_t mols_cap, pw_resident **res, int32_t *n_mol, int32_t *status);
int pw_context_device(pw_context *ctx);

/* ---- shape descriptors and circumcircles --------------------------------------
 * get_gyration_tensor / get_inertia_tensor / get_tensor_eigenvalues(sort=True) / calc_asphericity /
 * calc_acylidricity / calc_relative_shape_anisotropy (utilities.py:434-650) of every unit of a batch
 * in one launch.  The tensors are bit-identical to the reference's (numpy's summation orders are
 * reproduced, including the N x N broadcast of utilities.py:511-522); the eigenvalues come from a
 * Jacobi iteration instead of LAPACK dgeev and agree to a few ulps of the largest one. */
typedef struct pw_shape_out {
    double gyration[3][3];
    double inertia[3][3];
    double eigenvalues[3];             /* of the inertia tensor, descending */
    double asphericity;
    double acylidricity;
    double relative_shape_anisotropy;
} pw_shape_out;
int pw_shape_batch(pw_context *ctx, const pw_batch_in *in, pw_shape_out *out);
/* circumcircle(coordinates, atom_sets) (utilities.py:1653-1691) for one molecule: n_sets triples of
 * atom indices -> diameters (n_sets) and centres (n_sets x 3). */
int pw_circumcircle(pw_context *ctx, const double *xyz, int64_t n_atoms, const int32_t *atom_sets,
                    int64_t n_sets, double *diameter, double *centre);

/* Native DL_POLY HISTORY ingest (trajectory.py:647-766): see pw_history_* in
 * pywindow_amd/csrc/pw_history.cpp */
typedef struct pw_history pw_history;
int pw_history_open(const char *path, pw_history **h);
int64_t pw_history_frames(const pw_history *h);
int64_t pw_history_atoms(const pw_history *h);
int pw_history_keytrj(const pw_history *h);
int pw_history_imcon(const pw_history *h);
/* atom keys of frame 0, NUL-separated, into buf (returns bytes needed) */
int64_t pw_history_atom_keys(const pw_history *h, char *buf, int64_t buflen);
/* coordinates of frames [first, first+count) -> xyz[count][natoms][3]; lattice[count][9] may be NULL */
int pw_history_read(const pw_history *h, int64_t first, int64_t count, double *xyz, double *lattice);
/* nstep and tstep of the "timestep" record of one frame (reference: frame_info, trajectory.py:712-721; natms,
 * keytrj and imcon of the record are the file's: pw_history_atoms / _keytrj / _imcon) */
int pw_history_frame_info(const pw_history *h, int64_t frame, int64_t *nstep, double *tstep);
/* host threads the reader decodes with (PW_READER_THREADS or the hardware concurrency, at most 16; they are
 * started once per process and parked between calls) */
int pw_history_reader_threads(void);
/* Frames [first_frame, first_frame + count) decoded into `staging` (count x natoms x 3, normally the context's
 * page-locked buffer: pw_context_pinned) and appended to the streamed batch `res` (pw_resident_stream_begin) as units
 * first_unit.. WHILE the decoding goes on: the reader's threads take blocks of frames, one more thread appends the
 * finished prefix whenever it has grown by min_append frames (<= 0: 64).  The frame loop of
 * Trajectory._analysis_serial (trajectory.py:496-522: read a frame, analyse it, read the next) with the reading and
 * the analysis side by side.  legs_ms (may be NULL): [0] ms until the last frame was decoded, [1] from there until the
 * last append had returned.  Called without holding anything: the appends take the context's mutex one at a time. */
int pw_history_stream_read(const pw_history *h, int64_t first_frame, int64_t count, pw_context *ctx, pw_resident *res,
                           int64_t first_unit, double *staging, int64_t min_append, double *legs_ms);
void pw_history_close(pw_history *h);

#ifdef __cplusplus
}
#endif
#endif /* PYWINDOW_AMD_H */
