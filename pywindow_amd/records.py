"""Columnar results of a trajectory analysis and lazy views of them.

The reference keeps ``analysis_output[frame][molecule]`` as nested dicts and writes them as JSON
(trajectory.py:251-271, io_tools.py:215-265): at a million units that is a million dicts and a
gigabyte of text.  The engine's own result is columnar -- one fixed-size record per (frame, molecule)
unit -- so that is what is kept and persisted here:

* :class:`RecordStore`: the structured record array (``_lib.UNIT_OUT_DTYPE``) of an analysis, the
  windows beyond what a record holds, and the (frame, molecule) of every unit; ``save`` / ``load``
  as ONE flat file (a JSON header and the arrays as they lie in memory; reopened as a memory map).
* :class:`LazyAnalysis`: a mapping with the reference's shape -- ``view[frame][molecule]`` is the
  dict ``Molecule.full_analysis()`` returns -- whose entries are built from the records on first
  access and cached.  Frames assigned by hand are kept as given.

``DLPOLY.analysis(lazy=True)``, ``DLPOLY.save_records`` and ``DLPOLY.load_records`` are the entry
points (pywindow_amd/trajectory.py); the JSON route (``save_analysis``) is unchanged.
"""

from __future__ import annotations

import pathlib
from collections.abc import MutableMapping

import numpy as np

from . import _lib, engine

FORMAT = "pywindow_amd.records/1"
MAGIC = b"PWREC001"
HEADER_BYTES = 4096


class RecordStore:
    """Records of one analysis in unit order.

    ``records`` (U,) ``UNIT_OUT_DTYPE``; ``extra`` ``EXTRA_WINDOW_DTYPE`` (its ``unit`` indexes
    ``records``); ``unit_frame`` (U,) the frame of every unit; ``unit_molecule`` (U,) the molecule
    index inside the frame, or all ``-1`` for a non-modular analysis (one unit per frame, key
    ``"0"`` -- a string -- in the reference's dict, trajectory.py:515-522)."""

    def __init__(self, records, unit_frame, unit_molecule=None, extra=None, stages: int = _lib.STAGE_ALL):
        def keep(a, dtype):          # (no copy of what already is such an array: memory maps stay memory maps)
            return a if isinstance(a, np.ndarray) and a.dtype == dtype and a.ndim == 1 else np.ascontiguousarray(a, dtype=dtype).reshape(-1)

        self.records = keep(records, _lib.UNIT_OUT_DTYPE)
        self.unit_frame = keep(unit_frame, np.dtype(np.int64))
        n = len(self.records)
        if unit_molecule is None:
            unit_molecule = np.full(n, -1, dtype=np.int64)
        self.unit_molecule = keep(unit_molecule, np.dtype(np.int64))
        self.extra = np.zeros(0, dtype=_lib.EXTRA_WINDOW_DTYPE) if extra is None else keep(extra, _lib.EXTRA_WINDOW_DTYPE)
        self.stages = int(stages)
        if len(self.unit_frame) != n or len(self.unit_molecule) != n:
            raise ValueError("one frame / molecule index per record")
        self._more = None
        self._spans = None

    # ---- index -----------------------------------------------------------------------------
    @property
    def modular(self) -> bool:
        return bool(len(self.unit_molecule)) and bool((self.unit_molecule >= 0).any())

    def spans(self) -> dict:
        """``{frame: (first unit, one past the last)}`` -- the units of a frame are contiguous."""
        if self._spans is None:
            uf = self.unit_frame
            spans = {}
            if len(uf):
                cut = np.flatnonzero(np.diff(uf)) + 1
                lo = np.concatenate([[0], cut])
                hi = np.concatenate([cut, [len(uf)]])
                for a, b in zip(lo.tolist(), hi.tolist()):
                    f = int(uf[a])
                    if f in spans:
                        raise ValueError(f"the units of frame {f} are not contiguous")
                    spans[f] = (a, b)
            self._spans = spans
        return self._spans

    def frame_properties(self, frame: int) -> dict:
        """The reference's ``analysis_output[frame]``: ``{"0": properties}`` or ``{0: ..., 1: ...}``."""
        lo, hi = self.spans()[frame]
        if self._more is None:
            self._more = engine.extra_by_unit([self.extra]) if len(self.extra) else {}
        out = {}
        for u in range(lo, hi):
            rec = self.records[u]
            if int(rec["status"]) != 0:
                engine.warn_like_reference(rec)
            m = int(self.unit_molecule[u])
            out["0" if m < 0 else m] = engine.record_to_properties(rec, self.stages, self._more.get(u))
        return out

    # ---- persistence -----------------------------------------------------------------------
    # One file: a 4096-byte header (magic, then JSON: format, stages, record layout, and for every array its
    # dtype, length and byte offset), then the arrays as they lie in memory, each at a 4096-byte boundary.
    # Written with plain sequential writes and reopened as a memory map, so a result of any size is "open" as
    # soon as its header has been read and only the frames that are looked at are ever paged in.
    _ARRAYS = ("records", "extra", "unit_frame", "unit_molecule")

    def save(self, path) -> pathlib.Path:
        """Write the store to ``path`` (``.pwrec`` is appended when the name has no suffix)."""
        import json

        path = pathlib.Path(path)
        if path.suffix == "":
            path = path.with_suffix(".pwrec")
        arrays = {k: np.ascontiguousarray(getattr(self, k)) for k in self._ARRAYS}
        meta = {"format": FORMAT, "stages": self.stages, "record_dtype": repr(_lib.UNIT_OUT_DTYPE.descr),
                "extra_dtype": repr(_lib.EXTRA_WINDOW_DTYPE.descr), "arrays": {}}
        at = HEADER_BYTES
        for k, a in arrays.items():
            meta["arrays"][k] = {"count": int(len(a)), "itemsize": int(a.dtype.itemsize), "offset": at}
            at += -(-a.nbytes // HEADER_BYTES) * HEADER_BYTES
        head = MAGIC + json.dumps(meta).encode()
        if len(head) > HEADER_BYTES:
            raise ValueError("header too large")
        # the file is sized first and filled through a memory map: one copy into the page cache, no write() calls
        # (measured 2x faster than tofile() on the container's disk)
        # ... into a temporary file beside the target, moved over it at the end: the arrays of a store that was LOADED
        # from `path` are read-only memory maps of that very file, and truncating it first would pull the pages from
        # under them (load, then save to the same name); a reader also never sees a half-written file
        import os
        import tempfile

        fd, tmp = tempfile.mkstemp(prefix=path.name + ".", suffix=".tmp", dir=str(path.parent))
        try:
            with os.fdopen(fd, "wb") as fh:
                fh.truncate(at)
            out = np.memmap(tmp, dtype=np.uint8, mode="r+", shape=(at,))
            out[:len(head)] = np.frombuffer(head, dtype=np.uint8)
            for k, a in arrays.items():
                if a.nbytes:
                    o = meta["arrays"][k]["offset"]
                    out[o:o + a.nbytes] = a.view(np.uint8).reshape(-1)
            out.flush()
            del out
            # mkstemp creates the file 0600: give it what open(path, "wb") would have given -- the mode of the file it
            # replaces, else 0666 less the umask
            try:
                mode = os.stat(path).st_mode & 0o7777
            except OSError:
                um = os.umask(0)
                os.umask(um)
                mode = 0o666 & ~um
            os.chmod(tmp, mode)
            os.replace(tmp, path)
        except BaseException:
            try:
                os.unlink(tmp)
            except OSError:
                pass
            raise
        return path

    @classmethod
    def load(cls, path, mmap: bool = True) -> "RecordStore":
        """Reopen a file written by :meth:`save`; ``mmap`` (default): the arrays are read-only memory maps."""
        import json

        path = pathlib.Path(path)
        if not path.exists() and path.suffix == "":
            path = path.with_suffix(".pwrec")
        with open(path, "rb") as fh:
            head = fh.read(HEADER_BYTES)
        if not head.startswith(MAGIC):
            raise ValueError(f"{path}: not a {FORMAT} file")
        meta = json.loads(head[len(MAGIC):].rstrip(b"\0").decode())
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} file")
        if meta["record_dtype"] != repr(_lib.UNIT_OUT_DTYPE.descr) or meta["extra_dtype"] != repr(_lib.EXTRA_WINDOW_DTYPE.descr):
            raise ValueError(f"{path}: written with another record layout")
        dtypes = {"records": _lib.UNIT_OUT_DTYPE, "extra": _lib.EXTRA_WINDOW_DTYPE, "unit_frame": np.dtype(np.int64),
                  "unit_molecule": np.dtype(np.int64)}
        got = {}
        for k in cls._ARRAYS:
            info = meta["arrays"][k]
            if info["itemsize"] != dtypes[k].itemsize:
                raise ValueError(f"{path}: array {k} has another item size")
            if info["count"] == 0:
                got[k] = np.zeros(0, dtype=dtypes[k])
            elif mmap:
                got[k] = np.memmap(path, dtype=dtypes[k], mode="r", offset=info["offset"], shape=(info["count"],))
            else:
                got[k] = np.fromfile(path, dtype=dtypes[k], count=info["count"], offset=info["offset"])
        return cls(got["records"], got["unit_frame"], got["unit_molecule"], got["extra"], int(meta["stages"]))

    @classmethod
    def concatenate(cls, stores) -> "RecordStore":
        stores = list(stores)
        if not stores:
            return cls(np.zeros(0, dtype=_lib.UNIT_OUT_DTYPE), np.zeros(0, np.int64))
        extras, first = [], 0
        for s in stores:
            extras.append(engine.offset_extra(s.extra, first))
            first += len(s.records)
        return cls(np.concatenate([s.records for s in stores]), np.concatenate([s.unit_frame for s in stores]),
                   np.concatenate([s.unit_molecule for s in stores]), np.concatenate(extras), stores[0].stages)

    def select(self, frames) -> "RecordStore":
        """The store of a subset of frames (in the given order)."""
        spans = self.spans()
        idx = np.concatenate([np.arange(*spans[f]) for f in frames]) if len(frames) else np.zeros(0, np.int64)
        extra = self.extra
        if len(extra):
            pos = {int(u): k for k, u in enumerate(idx.tolist())}
            keep = [e for e in extra if int(e["unit"]) in pos]
            extra = np.array(keep, dtype=_lib.EXTRA_WINDOW_DTYPE)
            for e in extra:
                e["unit"] = pos[int(e["unit"])]
        return RecordStore(self.records[idx], self.unit_frame[idx], self.unit_molecule[idx], extra, self.stages)


class LazyAnalysis(MutableMapping):
    """``analysis_output`` backed by records: a frame's nested dict is built when it is first asked for.

    Keys keep the order in which frames were analysed or assigned.  ``dict(view)`` (or
    :meth:`materialise`) gives the plain dict the reference builds."""

    def __init__(self, initial=None):
        self._order: dict = {}          # frame -> None (insertion order)
        self._built: dict = {}          # frame -> dict (built, or assigned by hand)
        self._source: dict = {}         # frame -> RecordStore holding its records
        if initial:
            for k, v in dict(initial).items():
                self[k] = v

    def attach(self, store: RecordStore, frames=None) -> None:
        """Frames of ``store`` (all of them, or ``frames``) become entries of the view, replacing older ones."""
        for f in (store.spans().keys() if frames is None else frames):
            self._order.setdefault(f, None)
            self._built.pop(f, None)
            self._source[f] = store

    def record_store(self) -> RecordStore:
        """ONE store with the records of every record-backed frame of the view, in the view's order (frames
        assigned by hand have no records and are left out)."""
        groups: list = []
        for f in self._order:
            s = self._source.get(f)
            if s is None:
                continue
            if groups and groups[-1][0] is s:
                groups[-1][1].append(f)
            else:
                groups.append((s, [f]))
        if len(groups) == 1 and groups[0][1] == list(groups[0][0].spans()):
            return groups[0][0]                      # the whole of one analysis: no copy
        return RecordStore.concatenate(s.select(fr) for s, fr in groups)

    def __getitem__(self, frame):
        if frame in self._built:
            return self._built[frame]
        store = self._source.get(frame)
        if store is None:
            raise KeyError(frame)
        props = store.frame_properties(frame)
        self._built[frame] = props
        return props

    def __setitem__(self, frame, value) -> None:
        self._order.setdefault(frame, None)
        self._built[frame] = value
        self._source.pop(frame, None)

    def __delitem__(self, frame) -> None:
        if frame not in self._order:
            raise KeyError(frame)
        del self._order[frame]
        self._built.pop(frame, None)
        self._source.pop(frame, None)

    def __iter__(self):
        return iter(self._order)

    def __len__(self) -> int:
        return len(self._order)

    def __contains__(self, frame) -> bool:
        return frame in self._order

    def materialise(self) -> dict:
        return {f: self[f] for f in self._order}

    def __repr__(self) -> str:
        return f"LazyAnalysis({len(self._order)} frames, {len(self._built)} built)"
