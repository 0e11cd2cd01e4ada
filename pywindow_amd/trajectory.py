"""Batched DL_POLY trajectory driver (counterpart of the reference's
``DLPOLY`` / ``Trajectory.analysis``, trajectory.py:350-586, 620-766).

The reference walks frames one by one -- parse in Python, build a ``Molecule``,
call ``full_analysis()`` -- optionally fanned out over a ``multiprocessing.Pool``.
Here the selected frames are tokenised by the native mmap parser
(csrc/pw_history.cpp), force-field keys are deciphered once (atom order is
constant across a HISTORY file), all (frame, molecule) units go to the GPU in ONE
launch, and the fixed-size result records are scattered back into the same
nested ``analysis_output[frame][mol_id]`` dict.  With ``torch.distributed``
initialised (one process per GPU), frames shard across ranks with no exchange and
a single gather of the records over RCCL/xGMI returns them to rank 0.
"""

from __future__ import annotations

import ctypes
import pathlib
import sys
import time

import numpy as np

from . import _lib, engine
from .element_data import MASS, VDW, element_ids
from .molecular import MolecularSystem, decipher_atom_key
from .records import LazyAnalysis, RecordStore

#: most frames a modular analysis pushes through the device in one piece (see Trajectory._run_modular)
MODULAR_CHUNK = 8192
#: ... a long one in pieces of MODULAR_PIECE frames, MODULAR_GROUP pieces to a group: a group's pieces are re-assembled
#: one after the other with the device to themselves (each while the reader decodes the next), then analysed back to back
MODULAR_PIECE = 512
MODULAR_GROUP = 16
MODULAR_IN_FLIGHT = 2      # (until round 5: pieces analysed while the next was re-assembled; PW_MODULAR_GROUP=0 is that schedule)
#: a plain analysis goes through in ONE piece up to 2 x RUN_PIECE frames and in pieces of RUN_PIECE beyond (what
#: bounds the device memory of a very long trajectory; see DLPOLY._run for why not smaller)
RUN_PIECE = 16384
#: a piece of at least this many frames is STREAMED: its analysis is launched first and the reader feeds it (the
#: launch's teams wait for the frames they are handed) -- STREAM_CHUNKS appends per piece
STREAM_MIN = 256
STREAM_CHUNKS = 4
#: consecutive frames go through the native streamed read: an append whenever this many more frames are decoded
STREAM_APPEND = 64


class _FunctionError(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class _TrajectoryError(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


_IMCON = {0: "nonperiodic", 1: "cubic", 2: "orthorhombic", 3: "parallelepiped",
          4: "truncated octahedral", 5: "rhombic dodecahedral", 6: "x-y parallelogram",
          7: "hexagonal prism"}
_KEYTRJ = {0: "coordinates", 1: "coordinates and velocities",
           2: "coordinates, velocities and forces"}


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block of ``ceil(n/world)`` items for ``rank``."""
    per = -(-n_items // world) if world > 0 else n_items
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


class DLPOLY:
    """A DL_POLY HISTORY trajectory (reference trajectory.py:589-833)."""

    def __init__(self, filepath) -> None:
        self.filepath = pathlib.Path(filepath)
        self.system_id = self.filepath.name.split(".")[0]
        self.frames: dict = {}
        self.analysis_output: dict = {}
        self._stores: list = []                 # the records behind analysis_output, one RecordStore per analysis
        #: host-side legs of the latest analysis_records / analysis call, milliseconds (the analysis itself runs
        #: asynchronously: what the host sees of it is the wait in the download)
        self.last_timings: dict = {}
        L = _lib.load()
        h = ctypes.c_void_p()
        rc = L.pw_history_open(str(self.filepath).encode(), ctypes.byref(h))
        if rc != 0:
            raise _TrajectoryError(f"cannot open/parse HISTORY file {self.filepath} (code {rc})")
        self._h = h
        self.no_of_frames = int(L.pw_history_frames(h))
        self.no_of_atoms = int(L.pw_history_atoms(h))
        self.periodic_boundary = _IMCON.get(int(L.pw_history_imcon(h)), "unknown")
        self.content_type = _KEYTRJ.get(int(L.pw_history_keytrj(h)), "unknown")
        need = int(L.pw_history_atom_keys(h, None, 0))
        if need < 0:
            raise _TrajectoryError("malformed first frame in HISTORY file")
        buf = ctypes.create_string_buffer(max(need, 1))
        L.pw_history_atom_keys(h, buf, need)
        self.atom_ids = np.array([k.decode() for k in buf.raw[:need].split(b"\0")[:-1]])

    def __del__(self):  # pragma: no cover
        try:
            if getattr(self, "_h", None):
                _lib.load().pw_history_close(self._h)
                self._h = None
        except Exception:
            pass

    # ---- frame access ---------------------------------------------------------------
    def read_coordinates(self, first: int, count: int, lattice: np.ndarray | None = None, out: np.ndarray | None = None) -> np.ndarray:
        """(count, natoms, 3) float64, parsed natively; ``lattice`` (count, 3, 3) receives the
        lattice matrices in the reference's orientation (cell vectors as columns,
        trajectory.py:724-726).  ``out``: a C-contiguous float64 array of that shape to decode into."""
        if out is not None:
            if out.shape != (count, self.no_of_atoms, 3) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError("out must be a C-contiguous float64 array of shape (count, natoms, 3)")
            xyz = out
        else:
            xyz = np.empty((count, self.no_of_atoms, 3), dtype=np.float64)
        if count:
            rc = _lib.load().pw_history_read(self._h, first, count, xyz.ctypes.data,
                                             None if lattice is None else lattice.ctypes.data)
            if rc != 0:
                raise _TrajectoryError(f"cannot decode frames {first}..{first + count - 1} (code {rc})")
        return xyz

    @property
    def periodic(self) -> bool:
        return self.periodic_boundary in ("cubic", "orthorhombic", "parallelepiped")

    def _read_selected(self, frames: list[int], want_lattice: bool, out: np.ndarray | None = None):
        """Coordinates (and lattices) of arbitrary frames; contiguous runs are one native call, decoded in
        place.  ``out``: where the coordinates go (e.g. a context's page-locked staging buffer)."""
        coords = out if out is not None else np.empty((len(frames), self.no_of_atoms, 3))
        lattice = np.zeros((len(frames), 3, 3)) if want_lattice else None
        i = 0
        while i < len(frames):
            j = i
            while j + 1 < len(frames) and frames[j + 1] == frames[j] + 1:
                j += 1
            self.read_coordinates(frames[i], j - i + 1, None if lattice is None else lattice[i : j + 1], out=coords[i : j + 1])
            i = j + 1
        return coords, lattice

    def elements(self, swap_atoms: dict | None = None, forcefield: str | None = None) -> np.ndarray:
        """Elements for every atom: swap keys, then decipher (reference
        trajectory.py:245-248 -> molecular.py:710-796), done once per trajectory."""
        keys = [str(k) for k in self.atom_ids]
        if swap_atoms is not None:
            keys = [swap_atoms.get(k, k) for k in keys]
        if forcefield is not None:
            keys = [decipher_atom_key(k, forcefield) for k in keys]
        return np.array(keys)

    def get_frames(self, frames="all", swap_atoms=None, forcefield=None):
        """MolecularSystem objects for the requested frames (int / list / 'all')."""
        sel = self._select(frames)
        el = self.elements(swap_atoms, forcefield)
        out = {}
        L = _lib.load()
        imcon = int(L.pw_history_imcon(self._h))
        keytrj = int(L.pw_history_keytrj(self._h))
        for f in sel:
            lat = np.zeros((1, 3, 3)) if self.periodic else None
            xyz = self.read_coordinates(f, 1, lat)[0]
            nstep, tstep = ctypes.c_int64(0), ctypes.c_double(0.0)
            if L.pw_history_frame_info(self._h, f, ctypes.byref(nstep), ctypes.byref(tstep)) != 0:
                raise _TrajectoryError(f"cannot decode the timestep record of frame {f}")
            # the keys of the reference's decoded frame, in its order (trajectory.py:712-766)
            sysd = {"frame_info": {"nstep": int(nstep.value), "natms": self.no_of_atoms, "keytrj": keytrj,
                                   "imcon": imcon, "tstep": float(tstep.value)}}
            if lat is not None:
                from .rebuild import lattice_array_to_unit_cell

                sysd["lattice"] = lat[0]
                sysd["unit_cell"] = lattice_array_to_unit_cell(lat[0])
            sysd["atom_ids"] = self.atom_ids.copy()
            sysd["coordinates"] = xyz
            sysd["elements"] = el.copy()
            out[f] = MolecularSystem.load_system(sysd, f"{self.system_id}_{f}")
            self.frames[f] = out[f]
        if isinstance(frames, int):
            return out[frames]
        return out

    def _select(self, frames) -> list[int]:
        """Frame selection rules of reference trajectory.py:436-471 (a (start, stop)
        tuple -- always an error there -- is accepted here as a range)."""
        if isinstance(frames, (int, np.integer)):
            return [int(frames)]
        if isinstance(frames, list):
            if not all(isinstance(f, (int, np.integer)) for f in frames):
                raise _FunctionError("The list should be populated with integers only.")
            return [int(f) for f in frames]
        if isinstance(frames, tuple) and len(frames) == 2:
            return list(range(int(frames[0]), int(frames[1])))
        if isinstance(frames, str):
            if frames in ("all", "everything"):
                return list(range(self.no_of_frames))
            raise _FunctionError("Didn't recognise the keyword. (see manual)")
        raise _FunctionError("frames must be an int, a list of ints, a (start, stop) tuple or 'all'")

    def save_analysis(self, filepath=None, override: bool = False) -> None:
        """Dump ``analysis_output`` as JSON (reference trajectory.py:251-271 -> io_tools.py:215-265: arrays
        become lists, frame numbers string keys; ``.json`` is appended unless the file name contains it;
        an existing file is only replaced with ``override=True`` -- ``FileExistsError`` otherwise)."""
        import json

        path = pathlib.Path(filepath) if filepath is not None else pathlib.Path.cwd() / f"{self.system_id}_pywindow_analysis"
        if ".json" not in path.name:
            path = path.with_suffix(".json")
        if override is False and path.is_file():
            raise FileExistsError(f"The file {path} already exists. Use a different filepath, or set the 'override' to True.")

        def enc(obj):
            if isinstance(obj, np.ndarray):
                return obj.tolist()
            raise TypeError("Not serializable")

        # (dumps, not dump: one pass of the C encoder instead of the chunk-by-chunk Python iterator)
        out = self.analysis_output.materialise() if isinstance(self.analysis_output, LazyAnalysis) else self.analysis_output
        path.write_text(json.dumps(out, default=enc))

    # ---- columnar persistence and lazy views --------------------------------------
    def _keep(self, store: RecordStore, lazy: bool) -> None:
        """Results of an analysis enter ``analysis_output``: as the reference's nested dicts (default), or --
        ``lazy`` -- as a view that builds a frame's dict from the records when it is first asked for."""
        self._stores.append(store)              # (indexed only when save_records / analysis_store ask for it)
        if lazy or isinstance(self.analysis_output, LazyAnalysis):
            if not isinstance(self.analysis_output, LazyAnalysis):
                self.analysis_output = LazyAnalysis(self.analysis_output)
            self.analysis_output.attach(store)
            return
        if store.modular:
            for f in store.spans():
                self.analysis_output[f] = {}
            props = engine.records_to_properties(store.records, store.stages, extra=store.extra)
            for p, f, m in zip(props, store.unit_frame.tolist(), store.unit_molecule.tolist()):
                self.analysis_output[f][m] = p
        else:
            props = engine.records_to_properties(store.records, store.stages, extra=store.extra)
            for f, p in zip(store.unit_frame.tolist(), props):
                self.analysis_output[f] = {"0": p}

    def save_records(self, filepath=None, override: bool = False):
        """The analysis so far as ONE flat file: the structured record array, the (frame, molecule) of every
        unit, the windows beyond what a record holds (pywindow_amd/records.py: JSON header + raw arrays,
        reopened as a memory map).  The columnar counterpart of ``save_analysis`` (reference
        trajectory.py:251-271 writes JSON): 500 000 units are 350 MB, a fraction of a second to write and
        milliseconds to reopen.  Same ``override`` rule as ``save_analysis``."""
        path = pathlib.Path(filepath) if filepath is not None else pathlib.Path.cwd() / f"{self.system_id}_pywindow_records"
        if path.suffix == "":
            path = path.with_suffix(".pwrec")
        if override is False and path.is_file():
            raise FileExistsError(f"The file {path} already exists. Use a different filepath, or set the 'override' to True.")
        return self.analysis_store.save(path)

    def load_records(self, filepath) -> RecordStore:
        """Reopen what ``save_records`` wrote: ``analysis_output`` becomes a lazy view of the records
        (``analysis_output[frame][molecule]`` is built on first access), ``analysis_store`` returns them."""
        store = RecordStore.load(filepath)
        self._keep(store, lazy=True)
        return store

    @property
    def analysis_store(self) -> RecordStore:
        """The records behind ``analysis_output`` (every frame analysed or loaded so far, in order; a frame
        analysed again with ``override`` keeps its place and takes the later records)."""
        view = LazyAnalysis()
        for store in self._stores:
            view.attach(store)
        return view.record_store()

    def analysis_records(self, frames="all", swap_atoms=None, forcefield=None, device=None) -> np.ndarray:
        """Columnar results: the structured record array (``_lib.UNIT_OUT_DTYPE``) for the
        selected frames, without building per-frame dicts."""
        sel = self._select(frames)
        ids = element_ids(self.elements(swap_atoms, forcefield))
        return self._run(sel, VDW[ids], MASS[ids], device)

    # ---- the hot path -----------------------------------------------------------------------
    def analysis(self, frames="all", ncpus: int = 1, ncpus_analysis: int = 1, override: bool = False,
                 modular: bool = False, rebuild: bool = False, swap_atoms: dict | None = None,
                 forcefield: str | None = None, device: int | None = None, distributed: bool | None = None,
                 lazy: bool = False):
        """``full_analysis`` of every selected frame in one launch per GPU.

        ``lazy=True``: ``analysis_output`` becomes a view of the result records that builds a frame's dict when
        it is first asked for (pywindow_amd/records.py) instead of a dict per unit up front -- at 500 000 units the
        dicts cost as much as the analysis.  ``save_records`` / ``load_records`` persist the records either way.

        ``ncpus`` / ``ncpus_analysis`` are accepted for API compatibility and
        ignored (there is no CPU path).  Results land in ``analysis_output[frame]["0"]``
        exactly like the reference's non-modular branch (trajectory.py:515-522).  With
        ``modular=True`` every frame is first split into discrete molecules
        (``rebuild=True``: re-assembled through the periodic boundary) by
        ``pw_discrete_molecules`` -- all frames in one launch -- and the results are keyed by
        the molecule index ``0..k-1`` (trajectory.py:512-514).
        """
        del ncpus, ncpus_analysis
        if modular is True:
            return self._analysis_modular(frames, override, rebuild is True, swap_atoms, forcefield, device, distributed, lazy)
        sel = self._select(frames)
        if not override:
            sel = [f for f in sel if f not in self.analysis_output]
        if not sel:
            return
        el = self.elements(swap_atoms, forcefield)
        ids = element_ids(el)
        vdw, mass = VDW[ids], MASS[ids]
        dist, rank, world = _dist_state(distributed)
        lo, hi = shard_range(len(sel), rank, world)
        mine = sel[lo:hi]
        if dist is not None and world > 1 and dist.get_backend() == "nccl":
            # one process per GPU: the records go from this rank's result buffer straight into the
            # RCCL gather (device pointer -> tensor view, stream-ordered, no host copy on the way)
            recs = self._run_and_gather_on_device(mine, vdw, mass, device, len(sel), rank, world, dist)
        else:
            err = None
            try:
                recs = self._run(mine, vdw, mass, device)
            except Exception as exc:  # noqa: BLE001  (local to this rank: the others must hear of it before the gather)
                err = exc
            _all_ranks_ok(dist, world, device, err)
            if dist is not None and world > 1:
                recs = gather_records(recs, len(sel), rank, world, dist)
        extra = engine.offset_extra(getattr(self, "_extra", np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE)), lo)
        if dist is not None and world > 1:
            extra = gather_extra(extra, rank, world, dist, device)
        if rank != 0:
            return
        self._keep(RecordStore(recs, np.asarray(sel, np.int64), None, extra), lazy)

    def _run(self, frames: list[int], vdw, mass, device):
        """Records of the given frames (tokenised by the native reader's threads, 1.3 ms per 1000 frames)."""
        self._extra = np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE)   # windows beyond what a record holds, by unit
        if not frames:
            return np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE)
        n = len(frames)
        ctx = engine.context(device)
        per = n if n <= 2 * RUN_PIECE else RUN_PIECE
        # Decode into the context's page-locked buffer, one pooled device block, copies by DMA, three
        # asynchronous launches.  Cutting a trajectory into small pieces that overlap on the device was
        # measured and is SLOWER (MI355X, file to records: 1000 frames 5.05 ms in one piece, 7.0 ms in four;
        # 10 000 frames 21.8 against 22.7 ms): the persistent window teams of a piece keep their LDS until the
        # piece's slowest optimiser chain is done, so the teams of later pieces wait for a place instead of
        # working -- one launch hands finished units to whichever team is free.  Pieces only bound memory.
        inflight, parts, extras = [], [], []
        done = [0]

        def collect():
            res = inflight[0]           # (stays listed until it has been freed: a download that raises leaks nothing)
            try:
                extra = []
                t0 = time.perf_counter()
                parts.append(res.download_settled(extra))
                self.last_timings["wait_download_ms"] += 1e3 * (time.perf_counter() - t0)
                extras.extend(engine.offset_extra(e, done[0]) for e in extra)
                done[0] += res.n_units
            finally:
                inflight.pop(0)
                res.free()              # (its device block goes back to the context's cache)

        # the context's staging buffer, its "records fetched last" list and its capacities are one per context:
        # one trajectory at a time goes through (threads analysing on the same device take turns here)
        timing = self.last_timings = {"tokenise_ms": 0.0, "upload_ms": 0.0, "launch_ms": 0.0, "wait_download_ms": 0.0,
                                      "pieces": 0, "streamed": False, "reader_threads": int(_lib.load().pw_history_reader_threads())}
        clock = time.perf_counter
        with ctx.lock:
            try:
                for lo in range(0, n, per):
                    sel = frames[lo:lo + per]
                    timing["pieces"] += 1
                    res = None
                    # (the page-locked buffer FIRST: growing it waits for the device, which must not be waiting for it)
                    buf = ctx.pinned_array((len(sel), self.no_of_atoms, 3))
                    if ctx.device >= 0 and len(sel) >= STREAM_MIN and n <= per:
                        # one piece, streamed: launch first, then decode chunk after chunk into the page-locked buffer
                        # and append -- the chains of the first frames run while the reader still decodes the rest
                        try:
                            res = ctx.stream_begin(len(sel), vdw, mass)
                        except _lib.PwHipError:
                            res = None               # (a molecule beyond LDS: the plain upload below)
                    if res is not None:
                        inflight.append(res)
                        t0 = clock()
                        res.launch(_lib.STAGE_ALL)
                        timing["launch_ms"] += 1e3 * (clock() - t0)
                        if sel[-1] - sel[0] + 1 == len(sel) and all(b == a + 1 for a, b in zip(sel, sel[1:])):
                            # consecutive frames: decoded and appended side by side by the native reader (its threads
                            # decode blocks of frames, one more appends the finished prefix every STREAM_APPEND frames)
                            try:
                                dec_ms, tail_ms = res.append_from_history(self._h, sel[0], buf, STREAM_APPEND)
                            except _lib.PwHipError as exc:     # (a frame that cannot be decoded: the reader's error)
                                raise _TrajectoryError(f"cannot decode frames {sel[0]}..{sel[-1]} ({exc})") from exc
                            timing["tokenise_ms"] += dec_ms
                            timing["upload_ms"] += tail_ms
                        else:
                            step = max(32, -(-len(sel) // STREAM_CHUNKS))
                            for a in range(0, len(sel), step):
                                t0 = clock()
                                self._read_selected(sel[a:a + step], False, out=buf[a:a + step])
                                t1 = clock()
                                res.append(buf[a:a + step])
                                t2 = clock()
                                timing["tokenise_ms"] += 1e3 * (t1 - t0)
                                timing["upload_ms"] += 1e3 * (t2 - t1)
                        timing["streamed"] = True
                    else:
                        t0 = clock()
                        coords, _ = self._read_selected(sel, False, out=buf)
                        t1 = clock()
                        res = ctx.upload(_lib.Batch.uniform(coords, vdw, mass))
                        inflight.append(res)
                        t2 = clock()
                        res.launch(_lib.STAGE_ALL)
                        t3 = clock()
                        timing["tokenise_ms"] += 1e3 * (t1 - t0)
                        timing["upload_ms"] += 1e3 * (t2 - t1)
                        timing["launch_ms"] += 1e3 * (t3 - t2)
                    if len(inflight) > 2:   # at most three pieces on the device, however long the trajectory
                        collect()
                while inflight:
                    collect()
            finally:
                for res in inflight:
                    res.free()
        if extras:
            self._extra = np.concatenate(extras)
        recs = parts[0] if len(parts) == 1 else np.concatenate(parts)
        engine.raise_on_uncomputable(recs)
        return recs

    def _run_and_gather_on_device(self, frames, vdw, mass, device, n_total, rank, world, dist):
        import torch

        dev = engine.resolve_device(device)
        res = None
        ctx = engine.context(dev)
        with ctx.lock:
            err = None
            try:
                if frames:
                    coords, _ = self._read_selected(frames, False)
                    res = ctx.upload(_lib.Batch.uniform(coords, vdw, mass))
                    res.launch(_lib.STAGE_ALL)
            except Exception as exc:  # noqa: BLE001  (local to this rank: the others must hear of it before the gather)
                err = exc
            self._extra = np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE)
            try:
                _all_ranks_ok(dist, world, dev, err)
                recs = gather_records_device(res, n_total, rank, world, dist, torch.device("cuda", dev))
                if res is not None:
                    # the gather read the records on the device; what only the host can see comes now: a
                    # window launch that timed out (raises), and the windows beyond what a record holds
                    try:
                        self._extra = res.check()
                    except _lib.PwRetry:
                        res.launch(_lib.STAGE_ALL)
                        self._extra = res.check()
                engine.raise_on_uncomputable(recs)
                return recs
            finally:
                if res is not None:
                    res.free()

    # ---- modular analysis: frames -> discrete molecules -> units --------------------------------
    def modular_records(self, frames="all", rebuild: bool = False, swap_atoms=None, forcefield=None, device=None):
        """Columnar form of the modular analysis for the selected frames of THIS process:
        ``(records, unit_frame, unit_molecule)``; ``records[k]`` belongs to molecule
        ``unit_molecule[k]`` of frame ``unit_frame[k]``."""
        return self._run_modular(self._select(frames), rebuild, self.elements(swap_atoms, forcefield), device)

    def _run_modular(self, frames: list[int], rebuild: bool, el, device):
        from . import rebuild as rb

        if not frames:
            return np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE), np.zeros(0, np.int64), np.zeros(0, np.int64)
        if rebuild and not self.periodic:
            raise KeyError("lattice")   # create_supercell needs the cell (utilities.py:776-779)
        # (per-atom constants of the trajectory: the same for every call with the same elements)
        key = np.asarray(el).tobytes()
        cached = getattr(self, "_topology", None)
        if cached is None or cached[0] != key:
            ids = element_ids(el)
            cached = self._topology = (key, rb.CellTopology(el), np.ascontiguousarray(VDW[ids]))
        topo, vdw = cached[1], cached[2]
        dev = engine.resolve_device(device)

        # frames -> molecules -> units without leaving the device: every molecule of every frame of a
        # piece is one unit of ONE analysis launch.  Long trajectories go through in pieces (the frames
        # and the re-assembled molecules of a piece are resident on the device at once; each piece is two
        # launches), and the pieces in GROUPS: first every piece of a group is re-assembled -- the device to itself,
        # piece k on the device while the reader's threads decode piece k + 1 -- then the group's analyses are
        # launched back to back (different batches: nothing paces them) and collected.  Until round 5 the analysis of
        # piece k was launched at once and piece k + 1 re-assembled beside it: a re-assembly team needs 74 KB of LDS
        # and cannot be placed while the analysis' persistent teams hold the CUs -- the launch that takes 1.9 ms alone
        # (512 frames, copy included) took 6 beside an analysis, on a stream of the highest priority as on the API
        # stream, and the next analysis waited for it (profiles/r06_periodic_*: 1024 frames 19.6 -> 16 ms, 10 000 frames
        # 205 -> 125 ms).  PW_MODULAR_GROUP=0: the old schedule (MODULAR_IN_FLIGHT pieces waiting for their download).
        ctx = engine.context(dev)
        n = len(frames)
        import os as _os

        m_piece = int(_os.environ.get("PW_MODULAR_PIECE", MODULAR_PIECE))
        m_flight = int(_os.environ.get("PW_MODULAR_IN_FLIGHT", MODULAR_IN_FLIGHT))
        m_group = int(_os.environ.get("PW_MODULAR_GROUP", MODULAR_GROUP))
        piece = MODULAR_CHUNK if n < 2 * m_piece else min(MODULAR_CHUNK, m_piece)
        if -(-n // piece) < 4:
            m_group = 0          # (two or three pieces: the analysis of the first beside the re-assembly of the second wins)
        parts, waiting, spent = [], [], []
        # host-side legs, milliseconds: waiting for the reader (it runs on a helper thread beside the device work), the
        # re-assembly call (copies up, rebuild launch, its wait, the on-device hand-over), queueing the analysis,
        # waiting for records (the analysis of a piece overlaps the re-assembly of the next)
        timing = self.last_timings = {"tokenise_wait_ms": 0.0, "rebuild_ms": 0.0, "launch_ms": 0.0, "wait_download_ms": 0.0,
                                      "pieces": 0, "reader_threads": int(_lib.load().pw_history_reader_threads())}
        clock = time.perf_counter

        extras: list = []
        done = [0]

        def collect():
            res, n_mol = waiting[0]        # (stays listed until it has been freed: a download that raises leaks nothing)
            try:
                extra: list = []
                t0 = clock()
                recs = res.download_settled(extra) if res is not None else np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE)
                timing["wait_download_ms"] += 1e3 * (clock() - t0)
                extras.extend(engine.offset_extra(e, done[0]) for e in extra)
                done[0] += len(recs)
                parts.append((recs, n_mol))
            finally:
                waiting.pop(0)
                if res is not None:
                    res.free()             # (its blocks go back to the context's cache: no device-wide wait, and the
                                           # memory of a long trajectory stays at the pieces in flight)

        # Pieces are decoded into the context's page-locked buffer, two halves in turn (a piece has been copied to the
        # device when resident_from_cells returns, i.e. before the piece after next is decoded): the H2D of a
        # 1344-atom x 512-frame piece is then one DMA instead of a staged copy through the runtime's bounce buffer.
        n_pieces = -(-n // piece)
        halves = [None]                    # (taken under the context's lock, below: asking for a larger buffer frees
                                           # the one another thread on this context may be decoding into)

        def read_piece(i, k=0):
            sel_ = frames[i:i + piece]
            out = halves[0][k % len(halves[0])][: len(sel_)] if halves[0] is not None else None
            coords, lattice = self._read_selected(sel_, self.periodic, out=out)
            return rb.pack_frames(coords, lattice)

        # the next piece is tokenised by a helper thread (the native reader releases the interpreter lock)
        # while this thread waits for the re-assembly launch of the current one
        starts = list(range(0, n, piece))
        pool = None
        if len(starts) > 1:
            from concurrent.futures import ThreadPoolExecutor

            pool = ThreadPoolExecutor(max_workers=1)
        ctx.lock.acquire()                 # (one trajectory at a time per context, see _run)
        try:
            if ctx.device >= 0 and n_pieces > 0:
                halves[0] = ctx.pinned_array((2 if n_pieces > 1 else 1, min(piece, n), self.no_of_atoms, 3))
            ahead = pool.submit(read_piece, starts[0], 0) if pool else None
            for k, i in enumerate(starts):
                t0 = clock()
                if pool:
                    coords, lat, inv = ahead.result()
                    ahead = pool.submit(read_piece, starts[k + 1], k + 1) if k + 1 < len(starts) else None
                else:
                    coords, lat, inv = read_piece(i, k)
                t1 = clock()
                res, n_mol = ctx.resident_from_cells(topo, vdw, coords, lat, inv, rebuild)
                t2 = clock()
                waiting.append((res, n_mol))
                timing["tokenise_wait_ms"] += 1e3 * (t1 - t0)
                timing["rebuild_ms"] += 1e3 * (t2 - t1)
                timing["pieces"] += 1
                if m_group <= 0:
                    # (the schedule of rounds 1-5: analysed at once, the next piece re-assembled beside it)
                    if res is not None:
                        res.launch(_lib.STAGE_ALL)
                    timing["launch_ms"] += 1e3 * (clock() - t2)
                    if len(waiting) > m_flight:
                        collect()
                    continue
                if len(waiting) >= m_group or k + 1 == len(starts):
                    # the group is re-assembled: its analyses, back to back; they are collected before the next group's
                    # first re-assembly is queued (the reader goes on decoding the next piece meanwhile)
                    for r_, _ in waiting:
                        if r_ is not None:
                            r_.launch(_lib.STAGE_ALL)
                    timing["launch_ms"] += 1e3 * (clock() - t2)
                    timing["groups"] = timing.get("groups", 0) + 1
                    while waiting:
                        collect()
            while waiting:
                collect()
        finally:
            if pool is not None:
                pool.shutdown(wait=True)
            for res in spent + [w[0] for w in waiting]:
                if res is not None:
                    res.free()
            ctx.lock.release()
        # (the records as bytes: numpy copies a structured array field by field, a byte array with memcpy)
        recs = parts[0][0] if len(parts) == 1 else np.concatenate(
            [np.ascontiguousarray(p[0]).view(np.uint8).reshape(-1) for p in parts]).view(_lib.UNIT_OUT_DTYPE)
        engine.raise_on_uncomputable(recs)
        self._extra = np.concatenate(extras) if extras else np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE)
        n_mol = np.concatenate([p[1] for p in parts]).astype(np.int64)
        unit_frame = np.repeat(np.asarray(frames, np.int64), n_mol)
        # molecule index inside its frame: 0 .. n_mol[f] - 1 (position in the batch minus the frame's first unit)
        first = np.cumsum(n_mol) - n_mol
        unit_mol = np.arange(int(n_mol.sum()), dtype=np.int64) - np.repeat(first, n_mol)
        return recs, unit_frame, unit_mol

    def _analysis_modular(self, frames, override, rebuild, swap_atoms, forcefield, device, distributed, lazy=False):
        sel = self._select(frames)
        if not override:
            sel = [f for f in sel if f not in self.analysis_output]
        if not sel:
            return
        el = self.elements(swap_atoms, forcefield)
        dist, rank, world = _dist_state(distributed)
        lo, hi = shard_range(len(sel), rank, world)
        err = None
        try:
            recs, uframe, umol = self._run_modular(sel[lo:hi], rebuild, el, device)
        except Exception as exc:  # noqa: BLE001  (local to this rank)
            err = exc
        _all_ranks_ok(dist, world, device, err)
        extra = getattr(self, "_extra", np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE))
        if dist is not None and world > 1:
            tags = np.stack([uframe, umol], axis=1).astype(np.int64)
            dev = engine.resolve_device(device)
            # (units are numbered rank after rank in the gathered arrays)
            first = _exclusive_offset(np.array([len(recs)], np.int64), rank, world, dist, dev)
            extra = gather_extra(engine.offset_extra(extra, first), rank, world, dist, dev)
            recs = gather_ragged(recs, rank, world, dist, dev)
            tags = gather_ragged(tags.reshape(-1), rank, world, dist, dev)
            if rank != 0:
                return
            tags = tags.reshape(-1, 2)
            uframe, umol = tags[:, 0], tags[:, 1]
        store = RecordStore(recs, uframe, umol, extra)
        view = lazy or isinstance(self.analysis_output, LazyAnalysis)
        if not view:
            for f in sel:                      # (key order = selection order, as the reference's loop leaves it)
                self.analysis_output[f] = {}
        self._keep(store, lazy)
        if view:
            # a frame in which no molecule was found has no unit: the reference leaves an empty dict for it
            for f in sel:
                if f not in store.spans():
                    self.analysis_output[f] = {}


def _dist_state(distributed):
    """(torch.distributed module or None, rank, world) for an initialised process group."""
    if distributed is False:
        return None, 0, 1
    # a process group can only exist if the application imported torch.distributed already:
    # never pay the torch import (about a second) for a single-process analysis
    dist_mod = sys.modules.get("torch.distributed")
    if dist_mod is not None and dist_mod.is_available() and dist_mod.is_initialized():
        return dist_mod, dist_mod.get_rank(), dist_mod.get_world_size()
    return None, 0, 1


class PwRankError(RuntimeError):
    """Another rank of the job failed in a step that is local to a rank (reading, uploading, launching, downloading):
    every rank leaves the analysis together instead of waiting in a collective the failed rank will never enter."""


def _all_ranks_ok(dist, world: int, device, error: BaseException | None) -> None:
    """One all-reduce of a flag in front of a collective section.  ``error``: what this rank caught in its local
    step, or None.  Raises on EVERY rank when any rank failed: the rank's own exception where it has one,
    :class:`PwRankError` on the others.  (The counterpart in the reference: a worker's exception reaches the parent
    through ``pool.get()``, trajectory.py:563-586.)"""
    if dist is None or world <= 1:
        if error is not None:
            raise error
        return
    import torch

    flag = torch.tensor([1 if error is not None else 0], dtype=torch.int64, device=_collective_device(dist, device))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if error is not None:
        raise error
    if int(flag.item()) != 0:
        raise PwRankError("another rank failed in a local step of the analysis (see its output); no records were gathered")


def _collective_device(dist, device: int | None):
    """Where the tensors of a collective live: this rank's GPU for RCCL (``nccl``), the host for gloo."""
    import torch

    if dist.get_backend() == "nccl":
        return torch.device("cuda", engine.resolve_device(device))
    return torch.device("cpu")


def gather_ragged(local: np.ndarray, rank: int, world: int, dist, device: int | None = None) -> np.ndarray:
    """Concatenate per-rank arrays of different lengths on rank 0 (rank order): one
    ``all_gather`` of the lengths, one of the padded payloads."""
    import torch

    dev = _collective_device(dist, device)
    raw = np.ascontiguousarray(local).view(np.uint8).reshape(-1)
    size = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    buf = np.zeros(max(max(sizes), 1), dtype=np.uint8)
    buf[: raw.size] = raw
    send = torch.from_numpy(buf).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    if rank != 0:
        return np.zeros(0, dtype=local.dtype)
    parts = [recv[r].cpu().numpy()[: sizes[r]] for r in range(world)]
    return np.concatenate(parts).view(local.dtype)


def _exclusive_offset(count: np.ndarray, rank: int, world: int, dist, device) -> int:
    """Number of units on the ranks before this one (one all_gather of a counter)."""
    import torch

    dev = _collective_device(dist, device)
    mine = torch.tensor([int(count[0])], dtype=torch.int64, device=dev)
    alls = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(alls, mine)
    return int(sum(int(a.item()) for a in alls[:rank]))


def gather_extra(extra: np.ndarray, rank: int, world: int, dist, device=None) -> np.ndarray:
    """Windows beyond what a record holds, from every rank to rank 0 (``unit`` already numbered for the
    whole job).  Almost always there are none anywhere: one all-reduce of a flag says so and nothing
    else is exchanged."""
    import torch

    dev = _collective_device(dist, device)
    flag = torch.tensor([1 if len(extra) else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) == 0:
        return extra
    out = gather_ragged(extra, rank, world, dist, device)
    return out if rank == 0 else extra


def gather_records(local: np.ndarray, n_total: int, rank: int, world: int, dist, device: int | None = None) -> np.ndarray:
    """Host-record form of the path's only collective (used with gloo, and by callers that already
    hold their records on the host): every rank contributes its block of fixed-size result records,
    rank 0 receives all of them in frame order.

    Blocks are padded to ``ceil(n/world)`` records so a single equal-size
    ``all_gather`` suffices: a few hundred bytes per unit, once per trajectory.
    """
    import torch

    per = -(-n_total // world)
    rec_bytes = _lib.UNIT_OUT_DTYPE.itemsize
    buf = np.zeros(per * rec_bytes, dtype=np.uint8)
    raw = local.view(np.uint8).reshape(-1)
    buf[: raw.size] = raw
    dev = _collective_device(dist, device)
    send = torch.from_numpy(buf).to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    if rank != 0:
        return np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE)
    return _unpack_blocks([recv[r].cpu().numpy() for r in range(world)], n_total, world)


def _unpack_blocks(blocks, n_total: int, world: int) -> np.ndarray:
    rec_bytes = _lib.UNIT_OUT_DTYPE.itemsize
    out = np.zeros(n_total, dtype=_lib.UNIT_OUT_DTYPE)
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        if hi > lo:
            out[lo:hi] = np.ascontiguousarray(blocks[r][: (hi - lo) * rec_bytes]).view(_lib.UNIT_OUT_DTYPE)
    return out


class _DeviceBytes:
    """``__cuda_array_interface__`` over a raw device address: lets ``torch.as_tensor`` wrap the
    engine's result buffer without a copy."""

    def __init__(self, ptr: int, nbytes: int) -> None:
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_bytes_tensor(ptr: int, nbytes: int, device):
    """uint8 torch tensor that aliases ``nbytes`` of device memory at ``ptr`` (no copy)."""
    import torch

    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=device)


def gather_records_device(res, n_total: int, rank: int, world: int, dist, device, out=None):
    """The path's only collective, device to device: ``res`` (a ``_lib.Resident`` with a launch in
    flight, or ``None`` for a rank without frames) hands its record buffer to PyTorch's current stream
    (``pw_resident_results_ready``: a stream wait, no host synchronisation), RCCL's all-gather reads
    it in place over xGMI, and only rank 0 copies the gathered block to the host.  Returns the
    records in frame order on rank 0, an empty array elsewhere.  ``out``: optional preallocated
    (world * per * record) uint8 device tensor, for callers that gather every step."""
    import torch

    per = -(-n_total // world)
    rec_bytes = _lib.UNIT_OUT_DTYPE.itemsize
    stream = torch.cuda.current_stream(device)
    mine = 0 if res is None else res.n_units
    if mine == per:
        send = device_bytes_tensor(res.results_ready(stream.cuda_stream), per * rec_bytes, device)
    else:
        # a short (last) or empty block: pad on the device
        send = torch.zeros(per * rec_bytes, dtype=torch.uint8, device=device)
        if mine:
            send[: mine * rec_bytes].copy_(device_bytes_tensor(res.results_ready(stream.cuda_stream),
                                                                mine * rec_bytes, device))
    if out is None:
        out = torch.empty(world * per * rec_bytes, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, send)
    if res is not None:
        res.results_release(stream.cuda_stream)
    if rank != 0:
        return np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE)
    host = out.cpu().numpy().reshape(world, per * rec_bytes)
    return _unpack_blocks([host[r] for r in range(world)], n_total, world)
