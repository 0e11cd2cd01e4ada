"""ctypes binding of libpywindow_hip.so (C ABI in include/pywindow_amd.h).

There is no CPU fallback: importing this module only loads the shared library;
every compute call needs a HIP device and raises :class:`PwHipError` otherwise.
"""

from __future__ import annotations

import ctypes
import pathlib

import numpy as np

W_MAX = 16
P_MAX = 2048

STAGE_BASIC = 1
STAGE_AVG = 2
STAGE_OPT = 4
STAGE_WINDOWS = 8
STAGE_ALL = 15

ST_NEGATIVE_PORE = 1
ST_WINDOW_OVERFLOW = 2
ST_POINTS_OVERFLOW = 4
ST_WINDOW_DROPPED = 8
ST_WINDOW_NEGATIVE = 16
ST_Z_BOUNDS = 32
ST_TOO_FEW_POINTS = 64
ST_PATH_TOO_LONG = 128

E_RETRY = -6
E_HIP = -3
E_TIMEOUT = -7
DBSCAN_MAX = 8192

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = _PKG / "libpywindow_hip.so"


class PwHipError(RuntimeError):
    """The HIP engine is missing, has no device, or a call failed."""


class PwRetry(PwHipError):
    """``PW_E_RETRY``: a capacity was grown for this batch; launch the analysis again."""


class PwTimeoutError(PwHipError):
    """``PW_E_TIMEOUT``: a launch of the pipeline gave up waiting for another one; the records are incomplete."""


class BatchIn(ctypes.Structure):
    _fields_ = [
        ("n_units", ctypes.c_int64),
        ("atom_offset", ctypes.POINTER(ctypes.c_int64)),
        ("xyz", ctypes.POINTER(ctypes.c_double)),
        ("vdw", ctypes.POINTER(ctypes.c_double)),
        ("mass", ctypes.POINTER(ctypes.c_double)),
        ("template_atoms", ctypes.c_int64),
    ]


class Params(ctypes.Structure):
    """``pw_params``: knobs of find_windows / find_average_diameter (reference defaults)."""

    _fields_ = [
        ("adjust_windows", ctypes.c_double),
        ("adjust_average", ctypes.c_double),
        ("increment", ctypes.c_double),
        ("pore_opt", ctypes.c_int32),
        ("opt_flags", ctypes.c_int32),
        ("opt_x0", ctypes.c_double * 3),
        ("opt_lo", ctypes.c_double * 3),
        ("opt_hi", ctypes.c_double * 3),
        ("increment2", ctypes.c_double),
        ("z_lo", ctypes.c_double),
        ("z_hi", ctypes.c_double),
        ("lb_z", ctypes.c_int32),
        ("z_second_mini", ctypes.c_int32),
    ]

    def __init__(self, adjust_windows=1.0, adjust_average=1.0, increment=1.0, pore_opt=True, opt_start=None,
                 opt_bounds=None, increment2=0.1, z_bounds=None, lb_z=True, z_second_mini=False):
        """``opt_start``: (3,) start of opt_pore_diameter; ``opt_bounds``: three (lo, hi) pairs, ``None``
        for an open side (scipy.optimize.minimize's ``bounds`` convention).  ``increment2``,
        ``z_bounds`` (a (lo, hi) pair), ``lb_z``, ``z_second_mini``: window_analysis's keywords
        (utilities.py:1191-1200)."""
        flags = 0
        x0 = (ctypes.c_double * 3)(0.0, 0.0, 0.0)
        lo = (ctypes.c_double * 3)(-np.inf, -np.inf, -np.inf)
        hi = (ctypes.c_double * 3)(np.inf, np.inf, np.inf)
        if opt_start is not None:
            flags |= 1
            for k in range(3):
                x0[k] = float(opt_start[k])
        if opt_bounds is not None:
            flags |= 2
            for k in range(3):
                a, b = opt_bounds[k]
                lo[k] = -np.inf if a is None else float(a)
                hi[k] = np.inf if b is None else float(b)
        z_lo, z_hi = (None, None) if z_bounds is None else z_bounds
        super().__init__(float(adjust_windows), float(adjust_average), float(increment), 1 if pore_opt else 0,
                         flags, x0, lo, hi, float(increment2),
                         -np.inf if z_lo is None else float(z_lo), np.inf if z_hi is None else float(z_hi),
                         1 if lb_z else 0, 0 if z_second_mini is False else 1)


class CellIn(ctypes.Structure):
    """``pw_cell_in``: frames of one system for ``pw_discrete_molecules``."""

    _fields_ = [
        ("n_frames", ctypes.c_int64),
        ("n_atoms", ctypes.c_int32),
        ("rebuild", ctypes.c_int32),
        ("xyz", ctypes.c_void_p),
        ("lattice", ctypes.c_void_p),
        ("lattice_inv", ctypes.c_void_p),
        ("cov", ctypes.c_void_p),
        ("mass", ctypes.c_void_p),
        ("terminal", ctypes.c_void_p),
        ("max_dist", ctypes.c_double),
        ("tol", ctypes.c_double),
    ]


class CellOut(ctypes.Structure):
    """``pw_cell_out``: caller-allocated result arrays of ``pw_discrete_molecules``."""

    _fields_ = [
        ("atoms_cap", ctypes.c_int32),
        ("mols_cap", ctypes.c_int32),
        ("n_mol", ctypes.c_void_p),
        ("status", ctypes.c_void_p),
        ("mol_offset", ctypes.c_void_p),
        ("src_atom", ctypes.c_void_p),
        ("src_image", ctypes.c_void_p),
        ("xyz", ctypes.c_void_p),
    ]


RB_ATOMS_OVERFLOW = 4
RB_MOLS_OVERFLOW = 8

#: numpy mirror of ``pw_unit_out`` (natural C alignment)
UNIT_OUT_DTYPE = np.dtype(
    [
        ("n_atoms", np.int32),
        ("status", np.int32),
        ("mw", np.float64),
        ("com", np.float64, (3,)),
        ("maxd", np.float64),
        ("maxd_i", np.int32),
        ("maxd_j", np.int32),
        ("avg_d", np.float64),
        ("pore_d", np.float64),
        ("pore_atom", np.int32),
        ("pore_opt_atom", np.int32),
        ("pore_vol", np.float64),
        ("pore_opt_d", np.float64),
        ("pore_opt_c", np.float64, (3,)),
        ("pore_vol_opt", np.float64),
        ("n_windows", np.int32),
        ("n_clusters", np.int32),
        ("win_d", np.float64, (W_MAX,)),
        ("win_c", np.float64, (W_MAX, 3)),
        ("n_points", np.int32),
        ("n_points_avg", np.int32),
        ("n_survivors", np.int32),
        ("opt_nit", np.int32),
        ("opt_nfev", np.int32),
        ("opt_task", np.int32),
        ("opt_msg", np.int32),
        ("n_eval", np.int32),
        ("eps", np.float64),
        ("sphere_r", np.float64),
    ],
    align=True,
)

#: numpy mirror of ``pw_extra_window``: a window beyond the W_MAX a record holds
EXTRA_WINDOW_DTYPE = np.dtype(
    [("unit", np.int64), ("index", np.int32), ("reserved", np.int32), ("d", np.float64), ("c", np.float64, (3,))],
    align=True,
)

#: numpy mirror of ``pw_unit_debug`` (stage capture of find_windows, ``Context.analyse_debug``)
UNIT_DEBUG_DTYPE = np.dtype(
    [
        ("n_survivors", np.int32),
        ("n_clusters", np.int32),
        ("pass_idx", np.int32, (P_MAX,)),
        ("labels", np.int32, (P_MAX,)),
        ("gap2", np.float64, (P_MAX,)),
        ("win", np.float64, (W_MAX, 12)),
    ],
    align=True,
)
#: columns of ``UNIT_DEBUG_DTYPE["win"]``
DEBUG_WIN_COLS = ("vx", "vy", "vz", "angle_1", "angle_2", "new_z", "d0", "z_x", "xy_x", "xy_y", "diam", "n_eval")

#: numpy mirror of ``pw_shape_out``
SHAPE_OUT_DTYPE = np.dtype(
    [
        ("gyration", np.float64, (3, 3)),
        ("inertia", np.float64, (3, 3)),
        ("eigenvalues", np.float64, (3,)),
        ("asphericity", np.float64),
        ("acylidricity", np.float64),
        ("relative_shape_anisotropy", np.float64),
    ],
    align=True,
)

#: every symbol include/pywindow_amd.h declares (checked by the CPU test-suite)
EXPORTED_SYMBOLS = [
    "pw_device_count",
    "pw_version",
    "pw_last_error",
    "pw_context_create",
    "pw_context_destroy",
    "pw_context_host_threads",
    "pw_context_pinned",
    "pw_params_default",
    "pw_context_set_params",
    "pw_analysis_batch",
    "pw_context_extra_windows",
    "pw_context_point_capacity",
    "pw_context_reserve_points",
    "pw_context_pipelined",
    "pw_context_gate_timeouts",
    "pw_context_retries",
    "pw_context_count_retry",
    "pw_retries_total",
    "pw_context_queue_state",
    "pw_analysis_debug",
    "pw_point_gaps",
    "pw_pairwise_sum",
    "pw_dbscan",
    "pw_resident_upload",
    "pw_resident_stream_begin",
    "pw_resident_stream_append",
    "pw_resident_launch",
    "pw_resident_sync",
    "pw_resident_download",
    "pw_resident_free",
    "pw_resident_time",
    "pw_resident_stage_times",
    "pw_resident_device_results",
    "pw_resident_extra_windows",
    "pw_resident_results_ready",
    "pw_resident_results_release",
    "pw_resident_units",
    "pw_context_stream",
    "pw_context_device",
    "pw_discrete_molecules",
    "pw_resident_from_cells",
    "pw_shape_batch",
    "pw_circumcircle",
    "pw_history_open",
    "pw_history_frames",
    "pw_history_atoms",
    "pw_history_keytrj",
    "pw_history_imcon",
    "pw_history_atom_keys",
    "pw_history_read",
    "pw_history_frame_info",
    "pw_history_reader_threads",
    "pw_history_stream_read",
    "pw_history_close",
]

_lib = None


def _share_torch_hip_runtime() -> None:
    """One HIP / HSA runtime per process.  PyTorch-ROCm ships its own ``libamdhip64`` (same SONAME as
    the system one this library links to).  If torch is imported first the loader gives this library
    torch's copy; the other way round the process would end up with two runtimes, and the second
    (torch's) finds no GPU.  So when torch is installed its copy is loaded first, without importing
    torch.  ``PW_SYSTEM_HIP=1`` keeps the system runtime."""
    import importlib.util
    import os
    import sys

    if os.environ.get("PW_SYSTEM_HIP") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = pathlib.Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libpywindow_hip.so (built by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PwHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    # the pipeline runs several kernels side by side: give the runtime enough hardware queues
    import os

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    _share_torch_hip_runtime()
    try:
        L = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise PwHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp = ctypes.c_void_p
    L.pw_device_count.restype = ctypes.c_int
    L.pw_version.restype = ctypes.c_char_p
    L.pw_last_error.restype = ctypes.c_char_p
    L.pw_context_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.pw_context_host_threads.argtypes = [vp, ctypes.c_int]
    L.pw_context_pinned.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.pw_context_destroy.argtypes = [vp]
    L.pw_context_destroy.restype = None
    L.pw_params_default.argtypes = [ctypes.POINTER(Params)]
    L.pw_params_default.restype = None
    L.pw_context_set_params.argtypes = [vp, ctypes.POINTER(Params)]
    L.pw_analysis_batch.argtypes = [vp, ctypes.POINTER(BatchIn), ctypes.c_uint32, vp]
    L.pw_analysis_debug.argtypes = [vp, ctypes.POINTER(BatchIn), ctypes.c_uint32, vp, vp]
    L.pw_context_extra_windows.argtypes = [vp, vp, ctypes.c_int64]
    L.pw_context_extra_windows.restype = ctypes.c_int64
    L.pw_context_point_capacity.argtypes = [vp]
    L.pw_context_reserve_points.argtypes = [vp, ctypes.c_int64]
    L.pw_context_pipelined.argtypes = [vp]
    L.pw_context_gate_timeouts.argtypes = [vp, vp]
    L.pw_context_retries.argtypes = [vp, vp]
    L.pw_context_count_retry.argtypes = [vp]
    L.pw_retries_total.argtypes = []
    L.pw_retries_total.restype = ctypes.c_uint64
    L.pw_context_queue_state.argtypes = [vp, vp, ctypes.c_int]
    L.pw_point_gaps.argtypes = [vp, ctypes.POINTER(BatchIn), vp, vp, ctypes.c_int64, vp, vp]
    L.pw_pairwise_sum.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int, vp]
    L.pw_dbscan.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_double, ctypes.c_int, vp, vp]
    L.pw_resident_upload.argtypes = [vp, ctypes.POINTER(BatchIn), ctypes.POINTER(vp)]
    L.pw_resident_stream_begin.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.POINTER(vp)]
    L.pw_resident_stream_append.argtypes = [vp, vp, vp, ctypes.c_int64, ctypes.c_int64]
    L.pw_resident_launch.argtypes = [vp, vp, ctypes.c_uint32]
    L.pw_resident_sync.argtypes = [vp]
    L.pw_resident_download.argtypes = [vp, vp, vp]
    L.pw_resident_free.argtypes = [vp, vp]
    L.pw_resident_free.restype = None
    L.pw_resident_time.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.pw_resident_stage_times.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.pw_resident_extra_windows.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_int64)]
    L.pw_resident_device_results.argtypes = [vp]
    L.pw_resident_device_results.restype = vp
    L.pw_resident_results_ready.argtypes = [vp, vp, vp, ctypes.POINTER(vp)]
    L.pw_resident_results_release.argtypes = [vp, vp, vp]
    L.pw_resident_units.argtypes = [vp]
    L.pw_resident_units.restype = ctypes.c_int64
    L.pw_context_stream.argtypes = [vp]
    L.pw_context_stream.restype = vp
    L.pw_context_device.argtypes = [vp]
    L.pw_discrete_molecules.argtypes = [vp, ctypes.POINTER(CellIn), ctypes.POINTER(CellOut)]
    L.pw_resident_from_cells.argtypes = [vp, ctypes.POINTER(CellIn), vp, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.POINTER(vp), vp, vp]
    L.pw_shape_batch.argtypes = [vp, ctypes.POINTER(BatchIn), vp]
    L.pw_circumcircle.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]
    L.pw_history_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.pw_history_frames.argtypes = [vp]
    L.pw_history_frames.restype = ctypes.c_int64
    L.pw_history_atoms.argtypes = [vp]
    L.pw_history_atoms.restype = ctypes.c_int64
    L.pw_history_keytrj.argtypes = [vp]
    L.pw_history_imcon.argtypes = [vp]
    L.pw_history_atom_keys.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    L.pw_history_atom_keys.restype = ctypes.c_int64
    L.pw_history_read.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, vp]
    L.pw_history_frame_info.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    L.pw_history_reader_threads.argtypes = []
    L.pw_history_stream_read.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]
    L.pw_history_close.argtypes = [vp]
    L.pw_history_close.restype = None
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().pw_last_error().decode(errors="replace")
        if rc == E_TIMEOUT:
            raise PwTimeoutError(f"{what} failed with code {rc}: {msg}")
        raise PwHipError(f"{what} failed with code {rc}: {msg}")


def timeout_repeats() -> int:
    """How often an analysis is repeated after ``PW_E_TIMEOUT`` before the error is raised (``PW_TIMEOUT_REPEATS``,
    default 2; the library's ``pw_analysis_batch`` reads the same variable)."""
    import os

    try:
        return max(0, int(os.environ.get("PW_TIMEOUT_REPEATS", "2")))
    except ValueError:
        return 2


def retries_total() -> int:
    """Analyses repeated after a launch gave up waiting for another one (``PW_E_TIMEOUT``), over every context
    of this process.  Zero on a healthy device."""
    return int(load().pw_retries_total())


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class Batch:
    """Host-side description of a ragged batch of molecules (keeps arrays alive)."""

    def __init__(self, atom_offset, xyz, vdw, mass, template_atoms: int = 0):
        """``template_atoms`` = T > 0: every unit has T atoms and ``vdw`` / ``mass`` are one template
        of T entries (``pw_batch_in.template_atoms``); 0: one entry per atom of the batch."""
        self.atom_offset = np.ascontiguousarray(atom_offset, dtype=np.int64)
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.vdw = np.ascontiguousarray(vdw, dtype=np.float64)
        self.mass = np.ascontiguousarray(mass, dtype=np.float64)
        n_atoms = int(self.atom_offset[-1]) if len(self.atom_offset) else 0
        n_const = int(template_atoms) if template_atoms else n_atoms
        if len(self.xyz) != n_atoms or len(self.vdw) != n_const or len(self.mass) != n_const:
            raise ValueError("atom_offset does not match the per-atom arrays")
        self.n_units = len(self.atom_offset) - 1
        self.template_atoms = int(template_atoms)
        self.c = BatchIn(
            self.n_units,
            self.atom_offset.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
            _dptr(self.xyz),
            _dptr(self.vdw),
            _dptr(self.mass),
            self.template_atoms,
        )

    @classmethod
    def uniform(cls, coords, vdw, mass):
        """``coords`` (U, N, 3) of one molecule type; ``vdw``/``mass`` (N,): the constants of the
        trajectory travel once, as a template."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        u, n, _ = coords.shape
        off = np.arange(u + 1, dtype=np.int64) * n
        return cls(off, coords.reshape(-1, 3), vdw, mass, template_atoms=n)


class Context:
    """One GPU: stream + workspace.  ``device`` is the HIP ordinal; ``-1`` is the explicit host path
    (the same kernel source run by host threads, ``host_threads`` of them) -- never chosen implicitly."""

    def __init__(self, device: int = 0, host_threads: int = 0):
        L = load()
        h = ctypes.c_void_p()
        _check(L.pw_context_create(device, ctypes.byref(h)), "pw_context_create")
        self._h = h
        self.device = device
        if device < 0 and host_threads > 0:
            L.pw_context_host_threads(h, int(host_threads))
        # params live on the context between set and reset: one analysis with params at a time
        import threading

        #: held across the multi-call protocols that go through per-context state -- the page-locked staging
        #: buffer (pinned_array -> upload), "the records fetched last" (download -> extra_windows), the capacities
        #: a repeated launch relies on.  The C library serialises single calls on a context by itself
        #: (include/pywindow_amd.h, "Threads"); what belongs together is the caller's to keep together.
        self.lock = threading.RLock()
        self._params_lock = self.lock           # (params live on the context between set and reset: the same sequences)

    def close(self):
        if self._h:
            load().pw_context_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params: "Params | None" = None) -> None:
        """Knobs used by every later launch on this context (``None`` = reference defaults)."""
        p = params if params is not None else Params()
        _check(load().pw_context_set_params(self._h, ctypes.byref(p)), "pw_context_set_params")

    def extra_windows(self) -> np.ndarray:
        """Windows beyond ``W_MAX`` of the records fetched last on this context (``EXTRA_WINDOW_DTYPE``,
        ordered by unit and position): the reference has no limit on the number of windows."""
        n = load().pw_context_extra_windows(self._h, None, 0)
        buf = np.zeros(n, dtype=EXTRA_WINDOW_DTYPE)
        if n:
            load().pw_context_extra_windows(self._h, buf.ctypes.data, n)
        return buf

    def pinned_array(self, shape) -> np.ndarray:
        """float64 array of ``shape`` in the context's page-locked staging buffer (``pw_context_pinned``):
        decode frames into it and the upload copies by DMA.  ONE buffer per context: the array is valid
        until the next call; an upload has finished with it when it returns.  Host contexts: a plain array."""
        n = int(np.prod(shape))
        if self.device < 0 or n == 0:
            return np.empty(shape, dtype=np.float64)
        ptr = ctypes.c_void_p()
        _check(load().pw_context_pinned(self._h, n * 8, ctypes.byref(ptr)), "pw_context_pinned")
        buf = (ctypes.c_double * n).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)
        return arr

    @property
    def pipelined(self) -> bool:
        """Whether analyses run as the overlapped pipeline (needs ``GPU_MAX_HW_QUEUES`` >= 10 exported before
        the process first initialised HIP; measured at context creation) or as single launches."""
        return bool(load().pw_context_pipelined(self._h))

    @property
    def gate_timeouts(self) -> dict:
        """Diagnostic: pacing gates of the pipeline that gave up waiting since the context was created (a gate is
        never a dependency: 20 ms lost, no result changed).  All zero on a healthy device."""
        v = ctypes.c_uint64(0)
        _check(load().pw_context_gate_timeouts(self._h, ctypes.byref(v)), "pw_context_gate_timeouts")
        return {"tail": int(v.value & 0xffff), "head": int((v.value >> 16) & 0xffff), "residency": int(v.value >> 32)}

    @property
    def retries(self) -> int:
        """Analyses repeated on this context after ``PW_E_TIMEOUT`` (by the library or by this binding)."""
        v = ctypes.c_uint64(0)
        _check(load().pw_context_retries(self._h, ctypes.byref(v)), "pw_context_retries")
        return int(v.value)

    def queue_state(self) -> list:
        """Diagnostic: the hand-off queues of the pipeline's sets as they are now."""
        buf = (ctypes.c_uint64 * 16)()
        _check(load().pw_context_queue_state(self._h, buf, 16), "pw_context_queue_state")
        return [{"set": b, "taken": int(buf[4 * b]), "published": int(buf[4 * b + 1]), "started": int(buf[4 * b + 2]),
                 "error": int(buf[4 * b + 3])} for b in range(4)]

    def reserve_points(self, n_points: int) -> None:
        """At least ``n_points`` sampling vectors per molecule in the workspaces of every later launch
        (``pw_context_reserve_points``): what ``pw_analysis_batch`` does by itself when a unit asks for more
        than the ``adjust`` knobs imply, for callers of the resident entry points."""
        _check(load().pw_context_reserve_points(self._h, int(n_points)), "pw_context_reserve_points")

    @property
    def point_capacity(self) -> int:
        """Sampling vectors per molecule the workspaces hold at present (follows the ``adjust`` knobs)."""
        return int(load().pw_context_point_capacity(self._h))

    def analyse(self, batch: Batch, stages: int = STAGE_ALL, params: "Params | None" = None, extra=None) -> np.ndarray:
        """``extra``: a list that receives the ``EXTRA_WINDOW_DTYPE`` array of this analysis (windows
        beyond ``W_MAX``; read under the same lock as the analysis)."""
        out = np.zeros(batch.n_units, dtype=UNIT_OUT_DTYPE)
        if batch.n_units == 0:
            return out
        with self._params_lock:
            if params is not None:
                self.set_params(params)
            try:
                _check(
                    load().pw_analysis_batch(self._h, ctypes.byref(batch.c), stages, out.ctypes.data),
                    "pw_analysis_batch",
                )
                if extra is not None and (out["status"] & ST_WINDOW_OVERFLOW).any():
                    extra.append(self.extra_windows())
            finally:
                if params is not None:
                    self.set_params(None)
        return out

    def analyse_debug(self, batch: Batch, stages: int = STAGE_ALL):
        """``pw_analysis_debug``: ``(records, stage captures)`` -- the intermediate results of
        find_windows per unit (``UNIT_DEBUG_DTYPE``), for the parity tests."""
        out = np.zeros(batch.n_units, dtype=UNIT_OUT_DTYPE)
        dbg = np.zeros(batch.n_units, dtype=UNIT_DEBUG_DTYPE)
        if batch.n_units:
            with self._params_lock:
                _check(load().pw_analysis_debug(self._h, ctypes.byref(batch.c), stages, out.ctypes.data,
                                                dbg.ctypes.data), "pw_analysis_debug")
        return out, dbg

    def point_gaps(self, batch: Batch, unit_of_point, points):
        """min_i(|r_i - p| - vdw_i), argmin for each point (objective of the optimisers)."""
        u = np.ascontiguousarray(unit_of_point, dtype=np.int64)
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        gap = np.zeros(len(p))
        arg = np.zeros(len(p), dtype=np.int32)
        _check(
            load().pw_point_gaps(self._h, ctypes.byref(batch.c), u.ctypes.data, p.ctypes.data, len(p),
                                 gap.ctypes.data, arg.ctypes.data),
            "pw_point_gaps",
        )
        return gap, arg

    def dbscan(self, points, eps: float, one_wave: bool = False, global_memory: bool = False):
        """``DBSCAN(eps, min_samples=5).fit(points).labels_`` by one team on the GPU -> (labels, n_clusters)."""
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        labels = np.zeros(len(p), dtype=np.int32)
        k = ctypes.c_int32(0)
        _check(load().pw_dbscan(self._h, p.ctypes.data, len(p), float(eps), int(one_wave) | (int(global_memory) << 1),
                                labels.ctypes.data, ctypes.byref(k)), "pw_dbscan")
        return labels, k.value

    def pairwise_sum(self, values, one_wave: bool = False, global_scratch: bool = False) -> float:
        """``np.add.reduce`` of a float64 array in numpy's order, computed by one team on the GPU."""
        a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        out = ctypes.c_double(0.0)
        _check(load().pw_pairwise_sum(self._h, a.ctypes.data, len(a), int(one_wave) | (int(global_scratch) << 1),
                                      ctypes.byref(out)), "pw_pairwise_sum")
        return out.value

    def shape(self, batch: Batch) -> np.ndarray:
        """``pw_shape_batch``: gyration / inertia tensors, sorted eigenvalues and the three shape
        descriptors of every unit (``SHAPE_OUT_DTYPE`` records)."""
        out = np.zeros(batch.n_units, dtype=SHAPE_OUT_DTYPE)
        if batch.n_units:
            _check(load().pw_shape_batch(self._h, ctypes.byref(batch.c), out.ctypes.data), "pw_shape_batch")
        return out

    def circumcircle(self, coordinates, atom_sets):
        """``pw_circumcircle``: (diameters (K,), centres (K, 3)) for K atom triples of one molecule."""
        xyz = np.ascontiguousarray(coordinates, dtype=np.float64).reshape(-1, 3)
        sets = np.array(atom_sets, dtype=np.int32).reshape(-1, 3)
        sets = np.ascontiguousarray(np.where((sets < 0) & (sets >= -len(xyz)), sets + len(xyz), sets))   # Python indexing
        d = np.zeros(len(sets))
        c = np.zeros((len(sets), 3))
        rc = load().pw_circumcircle(self._h, xyz.ctypes.data, len(xyz), sets.ctypes.data, len(sets),
                                    d.ctypes.data, c.ctypes.data)
        if rc == -2 and len(sets) and ((sets < 0) | (sets >= len(xyz))).any():
            raise IndexError("atom index out of range")      # what indexing the array raises in the reference
        _check(rc, "pw_circumcircle")
        return d, c

    def discrete_molecules(self, topology, coords, lattice, lattice_inv, rebuild: bool, atoms_cap=None):
        """``pw_discrete_molecules`` on F frames of one topology (see pywindow_amd/rebuild.py).
        Returns ``(n_mol, mol_offset, src_atom, src_image, xyz, status)``; output capacity grows
        automatically when a frame reports an overflow."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        f, n, _ = coords.shape
        if n != topology.n:
            raise ValueError("coordinates do not match the topology")
        cap = int(atoms_cap) if atoms_cap else (2 * n if rebuild else n)
        mols = min(cap, n)
        for _attempt in range(6):
            n_mol = np.zeros(f, np.int32)
            status = np.zeros(f, np.int32)
            off = np.zeros((f, mols + 1), np.int32)
            src = np.zeros((f, cap), np.int32)
            img = np.zeros((f, cap), np.int8)
            xyz = np.zeros((f, cap, 3))
            cin = CellIn(f, n, 1 if rebuild else 0, coords.ctypes.data,
                         None if lattice is None else lattice.ctypes.data,
                         None if lattice_inv is None else lattice_inv.ctypes.data,
                         topology.cov.ctypes.data, topology.mass.ctypes.data, topology.terminal.ctypes.data,
                         topology.max_dist, topology.tol)
            cout = CellOut(cap, mols, n_mol.ctypes.data, status.ctypes.data, off.ctypes.data, src.ctypes.data,
                           img.ctypes.data, xyz.ctypes.data)
            _check(load().pw_discrete_molecules(self._h, ctypes.byref(cin), ctypes.byref(cout)),
                   "pw_discrete_molecules")
            if not (status & (RB_ATOMS_OVERFLOW | RB_MOLS_OVERFLOW)).any():
                break
            cap *= 4
            mols = min(cap, 4 * mols)
        else:
            raise PwHipError("pw_discrete_molecules: output does not fit (molecule larger than 2048 x the cell?)")
        bad = status & ~(RB_ATOMS_OVERFLOW | RB_MOLS_OVERFLOW)
        if bad.any():
            raise PwHipError(f"pw_discrete_molecules: unsupported input (status bits {int(np.bitwise_or.reduce(bad))}: "
                             "1/2 = more than 32 neighbours of one atom, 16 = cell thinner than the bond cut-off)")
        return n_mol, off, src, img, xyz

    def resident_from_cells(self, topology, vdw, coords, lattice, lattice_inv, rebuild: bool):
        """Frames -> discrete molecules -> ONE resident batch, all on the device
        (``pw_resident_from_cells``).  Returns ``(resident or None, n_mol per frame)``."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        vdw = np.ascontiguousarray(vdw, dtype=np.float64)
        f, n, _ = coords.shape
        if n != topology.n or len(vdw) != n:
            raise ValueError("coordinates / radii do not match the topology")
        cap = 2 * n if rebuild else n
        mols = min(cap, n)
        for _attempt in range(6):
            n_mol = np.zeros(f, np.int32)
            status = np.zeros(f, np.int32)
            cin = CellIn(f, n, 1 if rebuild else 0, coords.ctypes.data,
                         None if lattice is None else lattice.ctypes.data,
                         None if lattice_inv is None else lattice_inv.ctypes.data,
                         topology.cov.ctypes.data, topology.mass.ctypes.data, topology.terminal.ctypes.data,
                         topology.max_dist, topology.tol)
            h = ctypes.c_void_p()
            rc = load().pw_resident_from_cells(self._h, ctypes.byref(cin), vdw.ctypes.data, cap, mols,
                                               ctypes.byref(h), n_mol.ctypes.data, status.ctypes.data)
            if rc == -4:                 # PW_E_TOO_LARGE: a frame needs more room
                cap *= 4
                mols = min(cap, 4 * mols)
                continue
            _check(rc, "pw_resident_from_cells")
            break
        else:
            raise PwHipError("pw_resident_from_cells: output does not fit")
        bad = status & ~(RB_ATOMS_OVERFLOW | RB_MOLS_OVERFLOW)
        if bad.any():
            if h:
                load().pw_resident_free(self._h, h)
            raise PwHipError(f"pw_resident_from_cells: unsupported input (status bits {int(np.bitwise_or.reduce(bad))})")
        res = Resident._adopt(self, h, int(n_mol.sum())) if h else None
        return res, n_mol

    def upload(self, batch: Batch) -> "Resident":
        return Resident(self, batch)

    def stream_begin(self, n_units: int, vdw, mass) -> "Resident":
        """A batch of ``n_units`` molecules of one type whose coordinates will arrive in pieces
        (``pw_resident_stream_begin``): launch it at once, then ``append`` the coordinates in unit order."""
        vdw = np.ascontiguousarray(vdw, dtype=np.float64)
        mass = np.ascontiguousarray(mass, dtype=np.float64)
        if len(vdw) != len(mass) or not len(vdw):
            raise ValueError("one radius and one mass per atom of the molecule")
        h = ctypes.c_void_p()
        _check(load().pw_resident_stream_begin(self._h, int(n_units), len(vdw), vdw.ctypes.data, mass.ctypes.data,
                                               ctypes.byref(h)), "pw_resident_stream_begin")
        res = Resident._adopt(self, h, int(n_units))
        res.atoms = len(vdw)
        res.appended = 0
        return res

    @property
    def stream(self) -> int:
        return load().pw_context_stream(self._h) or 0


class Resident:
    """A batch kept in HBM across launches."""

    def __init__(self, ctx: Context, batch: Batch):
        self.ctx = ctx
        self.n_units = batch.n_units
        h = ctypes.c_void_p()
        _check(load().pw_resident_upload(ctx._h, ctypes.byref(batch.c), ctypes.byref(h)), "pw_resident_upload")
        self._h = h

    @classmethod
    def _adopt(cls, ctx: Context, handle, n_units: int) -> "Resident":
        obj = cls.__new__(cls)
        obj.ctx = ctx
        obj.n_units = n_units
        obj._h = handle
        return obj

    def append(self, coords) -> None:
        """Coordinates ``(count, atoms, 3)`` of the next units of a streamed batch (``Context.stream_begin``).
        ``coords`` should lie in the context's page-locked buffer (``Context.pinned_array``: the copy is then a DMA);
        the call returns when they are on the device."""
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        if coords.ndim != 3 or coords.shape[1:] != (self.atoms, 3):
            raise ValueError("coordinates must be (count, atoms, 3)")
        _check(load().pw_resident_stream_append(self.ctx._h, self._h, coords.ctypes.data, self.appended, len(coords)),
               "pw_resident_stream_append")
        self.appended += len(coords)

    def append_from_history(self, history_handle, first_frame: int, staging, min_append: int = 64) -> tuple[float, float]:
        """Frames ``first_frame ..`` of an open HISTORY (``pw_history*``) decoded into ``staging`` (``(count, atoms, 3)``,
        page-locked) and appended as the next units WHILE they are decoded (``pw_history_stream_read``).  Returns
        (ms until the last frame was decoded, ms from there until the last append had returned)."""
        if staging.ndim != 3 or staging.shape[1:] != (self.atoms, 3) or not staging.flags.c_contiguous or staging.dtype != np.float64:
            raise ValueError("staging must be a C-contiguous float64 (count, atoms, 3) array")
        legs = (ctypes.c_double * 2)()
        _check(load().pw_history_stream_read(history_handle, int(first_frame), len(staging), self.ctx._h, self._h, self.appended,
                                             staging.ctypes.data, int(min_append), legs), "pw_history_stream_read")
        self.appended += len(staging)
        return float(legs[0]), float(legs[1])

    def launch(self, stages: int = STAGE_ALL):
        self._stages = stages
        _check(load().pw_resident_launch(self.ctx._h, self._h, stages), "pw_resident_launch")

    def sync(self):
        _check(load().pw_resident_sync(self.ctx._h), "pw_resident_sync")

    def download(self, extra=None) -> np.ndarray:
        """Records of the latest launch.  ``extra``: a list that receives the windows beyond ``W_MAX``
        (``EXTRA_WINDOW_DTYPE``) when a unit has any.  The device list for those only exists once a launch
        has asked for it (``PW_E_RETRY``): the analysis is then launched once more, here."""
        out = np.zeros(self.n_units, dtype=UNIT_OUT_DTYPE)
        if self.n_units:
            with self.ctx.lock:       # (the records and "the windows beyond W_MAX of the records fetched last" belong together)
                rc = load().pw_resident_download(self.ctx._h, self._h, out.ctypes.data)
                if rc == E_RETRY:
                    self.launch(getattr(self, "_stages", STAGE_ALL))
                    rc = load().pw_resident_download(self.ctx._h, self._h, out.ctypes.data)
                repeats = 0
                while rc == E_TIMEOUT and repeats < timeout_repeats():
                    # a launch that saw another launch of the same analysis make no progress for a whole limit
                    # (PW_WAIT_LIMIT_MS, 250 ms): the analysis is repeated, up to PW_TIMEOUT_REPEATS (2) times -- the
                    # next time-out is raised -- and every repeat is COUNTED (Context.retries, retries_total(): the
                    # bench line and the suite's last test look at them) and logged
                    import logging

                    logging.getLogger("pywindow_amd").warning("analysis repeated after: %s", load().pw_last_error().decode(errors="replace"))
                    load().pw_context_count_retry(self.ctx._h)
                    repeats += 1
                    self.launch(getattr(self, "_stages", STAGE_ALL))
                    rc = load().pw_resident_download(self.ctx._h, self._h, out.ctypes.data)
                _check(rc, "pw_resident_download")
                if extra is not None and (out["status"] & ST_WINDOW_OVERFLOW).any():
                    extra.append(self.ctx.extra_windows())
        return out

    def download_settled(self, extra=None) -> np.ndarray:
        """``download`` for the resident path with what ``pw_analysis_batch`` does for the one-call path: a
        unit that wanted more sampling vectors than the launch's workspace held (``PW_ST_POINTS_OVERFLOW``:
        a sphere of thousands of angstroms) raises the context's capacity and the analysis is launched again,
        so no capacity of the engine shows in a result.  Download and extra windows under the context's lock."""
        with self.ctx.lock:
            attempt = 0
            while True:
                mine = []
                out = self.download(mine)
                flagged = out[(out["status"] & ST_POINTS_OVERFLOW) != 0]
                want = int(max(flagged["n_points"].max(), flagged["n_points_avg"].max())) if len(flagged) else 0
                attempt += 1
                # (no launch after the last download: what is returned is what the latest launch wrote)
                if want <= self.ctx.point_capacity or attempt >= 3:
                    break
                self.ctx.reserve_points(want)
                self.launch(getattr(self, "_stages", STAGE_ALL))
            if extra is not None:
                extra.extend(mine)
            return out

    def check(self) -> np.ndarray:
        """For callers that read the records on the device: wait for the latest launch, raise if its
        window launch timed out, and return its windows beyond ``W_MAX`` (``EXTRA_WINDOW_DTYPE``; empty
        for all but unusual molecules).  ``PW_E_RETRY`` (the device list for such windows had to be
        allocated first) is passed on as :class:`PwRetry`: launch again."""
        n = ctypes.c_int64(0)
        rc = load().pw_resident_extra_windows(self.ctx._h, self._h, ctypes.byref(n))
        if rc == E_RETRY:
            raise PwRetry(load().pw_last_error().decode(errors="replace"))
        _check(rc, "pw_resident_extra_windows")
        return self.ctx.extra_windows() if n.value else np.zeros(0, dtype=EXTRA_WINDOW_DTYPE)

    def time_launches(self, iters: int, stages: int = STAGE_ALL) -> float:
        ms = ctypes.c_float()
        _check(load().pw_resident_time(self.ctx._h, self._h, stages, iters, ctypes.byref(ms)), "pw_resident_time")
        return float(ms.value)

    def stage_times(self) -> dict:
        """Milliseconds of the launches of ONE analysis run on its own (HIP events on each launch's stream):
        ``{"chains", "windows"}`` -- the average diameter is a stage of the window teams -- or, with ``PW_B_LAUNCH=1``,
        ``{"chains", "average", "windows"}`` (the three launches of rounds 1-5)."""
        ms = (ctypes.c_float * 3)()
        _check(load().pw_resident_stage_times(self.ctx._h, self._h, ms), "pw_resident_stage_times")
        out = {"chains": float(ms[0]), "windows": float(ms[2])}
        if ms[1] > 0.0:
            out["average"] = float(ms[1])
        return out

    @property
    def device_results_ptr(self) -> int:
        return load().pw_resident_device_results(self._h) or 0

    def results_ready(self, stream: int = 0) -> int:
        """Make HIP stream ``stream`` (a ``hipStream_t`` as an integer; 0 = the default stream) wait
        for the latest launch; returns the device address of its records."""
        ptr = ctypes.c_void_p()
        _check(load().pw_resident_results_ready(self.ctx._h, self._h, ctypes.c_void_p(stream or None),
                                                ctypes.byref(ptr)), "pw_resident_results_ready")
        return ptr.value or 0

    def results_release(self, stream: int = 0) -> None:
        """The launch that next overwrites the records waits for what ``stream`` holds so far."""
        _check(load().pw_resident_results_release(self.ctx._h, self._h, ctypes.c_void_p(stream or None)),
               "pw_resident_results_release")

    def free(self):
        if self._h:
            load().pw_resident_free(self.ctx._h, self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.free()
        except Exception:
            pass
