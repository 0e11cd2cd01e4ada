"""Native DL_POLY HISTORY ingest and the host logic of the batched driver (no GPU)."""
import ctypes

import numpy as np
import pytest

from _util import load_group
from pywindow_amd import _lib, synth
from pywindow_amd.molecular import MolecularSystem, _AtomKeyConflictError, decipher_atom_key
from pywindow_amd.trajectory import DLPOLY, _FunctionError, shard_range


def test_history_roundtrip(tmp_path):
    path = synth.write_synthetic_history(tmp_path / "HISTORY", 7)
    traj = DLPOLY(path)
    assert (traj.no_of_frames, traj.no_of_atoms) == (7, 168)
    assert traj.periodic_boundary == "nonperiodic" and traj.content_type == "coordinates"
    el, frames = synth.synthetic_units(7)
    assert np.array_equal(traj.read_coordinates(0, 7), frames)
    assert np.array_equal(traj.read_coordinates(3, 2), frames[3:5])
    assert list(traj.elements()) == list(el)
    # the synthetic frames are the golden synth64 inputs
    g = load_group("synth64")
    assert np.array_equal(frames.reshape(-1, 3), g["coordinates"][: 7 * 168])
    ms = traj.get_frames(2)
    assert isinstance(ms, MolecularSystem) and np.array_equal(ms.system["coordinates"], frames[2])


def test_streamed_read_rejects_what_it_cannot_do(tmp_path):
    """pw_history_stream_read without a GPU: its argument checks (a null context or batch, frames beyond the file)
    -- the call itself is exercised by tests/test_gpu_api.py::test_streamed_batch_equals_the_uploaded_one."""
    path = synth.write_synthetic_history(tmp_path / "HISTORY", 5)
    traj = DLPOLY(path)
    L = _lib.load()
    buf = np.zeros((5, 168, 3))
    for args in ((traj._h, 0, 5, None, None, 0, buf.ctypes.data, 64, None),
                 (None, 0, 5, None, None, 0, buf.ctypes.data, 64, None),
                 (traj._h, 3, 5, None, None, 0, buf.ctypes.data, 64, None)):
        assert L.pw_history_stream_read(*args) != 0


def test_history_with_lattice_and_velocities(tmp_path):
    lines = ["title", "%10d%10d%10d" % (1, 3, 2)]
    for f in range(2):
        lines.append("timestep%10d%10d%10d%10d%12.6f" % (f, 2, 1, 3, 0.001))
        lines += ["  10.0 0.0 0.0", "  0.0 11.0 0.0", "  0.0 0.5 12.0"]
        for a in range(2):
            lines.append("%-8s%10d%12.6f%12.6f" % ("C" + str(a + 1), a + 1, 12.0, 0.0))
            lines.append("%12.4E%12.4E%12.4E" % (a + f, 2 * a, 3.5))
            lines.append("%12.4E%12.4E%12.4E" % (9, 9, 9))
    p = tmp_path / "H2"
    p.write_text("\n".join(lines) + "\n")
    L = _lib.load()
    h = ctypes.c_void_p()
    assert L.pw_history_open(str(p).encode(), ctypes.byref(h)) == 0
    assert (L.pw_history_frames(h), L.pw_history_atoms(h), L.pw_history_keytrj(h), L.pw_history_imcon(h)) == (2, 2, 1, 3)
    xyz = np.zeros((2, 2, 3))
    lat = np.zeros((2, 9))
    assert L.pw_history_read(h, 0, 2, xyz.ctypes.data, lat.ctypes.data) == 0
    assert np.array_equal(xyz[1], [[1.0, 0.0, 3.5], [2.0, 2.0, 3.5]])
    assert np.array_equal(lat[0].reshape(3, 3), np.array([[10, 0, 0], [0, 11, 0], [0, 0.5, 12]]).T)
    assert L.pw_history_read(h, 1, 2, xyz.ctypes.data, None) != 0   # out of range -> error code
    L.pw_history_close(h)
    assert L.pw_history_open(b"/nonexistent/HISTORY", ctypes.byref(h)) != 0


def test_force_field_keys():
    assert decipher_atom_key("ca", "opls") == "C" and decipher_atom_key("ni", "OPLS") == "N"
    assert decipher_atom_key("C12", "DLF") == "C" and decipher_atom_key("Zn1", "dl_f") == "Zn"
    with pytest.raises(_AtomKeyConflictError):
        decipher_atom_key("he", "opls")
    ms = MolecularSystem.load_system({"atom_ids": np.array(["he", "ca"]), "coordinates": np.zeros((2, 3))})
    ms.swap_atom_keys({"he": "H"})
    ms.decipher_atom_keys("opls")
    assert list(ms.system["elements"]) == ["H", "C"]


def test_frame_selection_and_sharding(tmp_path):
    traj = DLPOLY(synth.write_synthetic_history(tmp_path / "HISTORY", 5))
    assert traj._select("all") == [0, 1, 2, 3, 4] and traj._select(3) == [3] and traj._select((1, 3)) == [1, 2]
    with pytest.raises(_FunctionError):
        traj._select("bogus")
    with pytest.raises(_FunctionError):
        traj._select([0, "1"])
    # contiguous, exhaustive, ordered shards
    for n, w in ((1000, 8), (10, 3), (5, 8), (0, 2)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))


def test_number_decoding_equals_python_float(tmp_path):
    """The reference decodes every field with float(); the parser's fast path (short mantissa x
    exact power of ten) and its strtod fallback must give the same double for every spelling,
    and the threaded frame decode must keep the frame order."""
    rng = np.random.default_rng(17)
    toks = ["0", "-0.0", "+1.5", ".5", "5.", "1e5", "1E-5", "-2.5000E+00", "1.2345E+01",
            "9007199254740993", "9007199254740992.0", "0.1", "1e22", "1e23", "1e-22", "1e-23",
            "123456789012345678", "1234567890123456789012", "0.000000000000000000000001",
            "4.9406564584124654e-324", "1.7976931348623157e308", "2.2250738585072011e-308",
            "3.141592653589793238462643", "-7.0E-01", "6.02214076E+23", "1e+0", "00012.5e-3"]
    toks += ["%.4E" % v for v in rng.normal(0, 30, 40)]
    toks += ["%.17g" % v for v in rng.normal(0, 1e-3, 40)]
    toks += ["%.10f" % v for v in rng.normal(0, 1e3, 40)]
    while len(toks) % 3:
        toks.append("1.0")
    natms, nframes = len(toks) // 3, 40
    lines = ["title", "%10d%10d%10d" % (0, 0, natms)]
    for k in range(nframes):
        lines.append("timestep%10d%10d%10d%10d%12.6f" % (k + 1, natms, 0, 0, 0.001))
        for i in range(natms):
            lines.append("%-8s%10d%12.6f%12.6f" % ("C", i + 1, 12.0, 0.0))
            a, b, c = toks[3 * i: 3 * i + 3]
            # a different spelling of frame k's first field marks the frame
            lines.append("%s %s %s" % (a if i else "%d.0" % k, b, c))
    path = tmp_path / "HISTORY"
    path.write_text("\n".join(lines) + "\n")
    got = DLPOLY(path).read_coordinates(0, nframes)
    want = np.array([float(t) for t in toks]).reshape(natms, 3)
    for k in range(nframes):
        want[0, 0] = float(k)
        assert got[k].tobytes() == want.tobytes(), k


def test_bulk_record_conversion_equals_the_per_record_one():
    """engine.records_to_properties (column-wise) builds the same nested dicts, value for value and
    type for type, as record_to_properties does record by record."""
    from pywindow_amd import engine

    g = load_group("synth64")
    recs = np.zeros(6, dtype=_lib.UNIT_OUT_DTYPE)
    rng = np.random.default_rng(2)
    for name in recs.dtype.names:
        col = recs[name]
        if col.dtype.kind == "f":
            recs[name] = rng.normal(size=col.shape)
        else:
            recs[name] = rng.integers(0, 5, size=col.shape)
    recs["status"] = 0
    recs["n_windows"] = [4, 0, -1, 16, 2, 1]
    del g

    def same(a, b):
        assert type(a) is type(b), (type(a), type(b))
        if isinstance(a, dict):
            assert list(a) == list(b)
            for k in a:
                same(a[k], b[k])
        elif isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        else:
            assert a == b

    for stages in (_lib.STAGE_ALL, _lib.STAGE_BASIC, _lib.STAGE_BASIC | _lib.STAGE_AVG, _lib.STAGE_OPT):
        bulk = engine.records_to_properties(recs, stages)
        for rec, props in zip(recs, bulk):
            same(props, engine.record_to_properties(rec, stages))


def test_real_md_trajectory_through_the_native_reader(tmp_path):
    """The reference's own 20-frame DL_POLY data file (examples/data/input/HISTORY_singlemol_short, kept
    byte for byte in tests/golden/history20.npz) through pw_history_*: frame count, the real OPLS key
    strings, the elements after swap + decipher and every coordinate equal what the reference's
    reader produced (trajectory.py:217-249, 647-766; generated by tests/golden/make_golden.py)."""
    g = np.load(_util_golden() / "history20.npz")
    path = tmp_path / "HISTORY_singlemol_short"
    path.write_bytes(g["file_bytes"].tobytes())
    traj = DLPOLY(path)
    assert traj.no_of_frames == int(g["no_of_frames"]) == 20
    assert traj.no_of_atoms == 168 and traj.periodic_boundary == "nonperiodic"
    assert list(traj.atom_ids) == list(g["atom_ids"])
    assert "he" in set(traj.atom_ids)                      # the conflicting OPLS key the example swaps
    with pytest.raises(_AtomKeyConflictError):
        traj.elements(forcefield="opls")
    assert list(traj.elements(swap_atoms={"he": "H"}, forcefield="opls")) == list(g["elements"])
    assert np.array_equal(traj.read_coordinates(0, 20), g["coordinates"])
    assert np.array_equal(traj.read_coordinates(7, 3), g["coordinates"][7:10])
    # ... and they are the inputs of the md20 golden group
    md = load_group("md20")
    assert np.array_equal(g["coordinates"].reshape(-1, 3), md["coordinates"])
    assert list(np.tile(g["elements"], 20)) == list(md["elements"])


def _util_golden():
    from _util import GOLDEN

    return GOLDEN


def test_element_tables_equal_the_reference():
    """tables.py:22-286 (mass, van der Waals and covalent radii, 85 keys each, 'X' included) and the
    OPLS key table (tables.py:290-640) against the checksum fixture generated from the reference."""
    import hashlib

    from pywindow_amd import element_data as E

    g = np.load(_util_golden() / "tables.npz")
    keys = sorted(E.atomic_mass)
    assert keys == list(g["symbols"]) and len(keys) == 85 and "X" in keys
    assert keys == sorted(E.atomic_vdw_radius) == sorted(E.atomic_covalent_radius)
    mass = [E.atomic_mass[k] for k in keys]
    vdw = [E.atomic_vdw_radius[k] for k in keys]
    cov = [E.atomic_covalent_radius[k] for k in keys]
    assert np.array_equal(mass, g["mass"]) and np.array_equal(vdw, g["vdw"]) and np.array_equal(cov, g["covalent"])
    ids = E.element_ids(keys)
    assert np.array_equal(E.MASS[ids], g["mass"]) and np.array_equal(E.VDW[ids], g["vdw"])
    assert np.array_equal(E.COVALENT[ids], g["covalent"])
    opls = sorted(E.OPLS_KEY_TO_ELEMENT)
    assert opls == list(g["opls_keys"])
    assert [E.OPLS_KEY_TO_ELEMENT[k] for k in opls] == list(g["opls_elements"])
    text = "\n".join(f"{k} {m!r} {v!r} {c!r}" for k, m, v, c in zip(keys, mass, vdw, cov))
    text += "\n" + "\n".join(f"{k} {E.OPLS_KEY_TO_ELEMENT[k]}" for k in opls)
    assert hashlib.sha256(text.encode()).hexdigest() == str(g["sha256"])
    assert (E.atomic_mass["C"], E.atomic_vdw_radius["C"]) == (12.011, 1.7)
    assert (E.atomic_mass["H"], E.atomic_vdw_radius["H"]) == (1.008, 1.09)


def test_save_analysis_override_semantics(tmp_path):
    """Trajectory.save_analysis (reference trajectory.py:251-271 -> io_tools.py:215-265): '.json' is
    appended unless the name contains it, an existing file raises FileExistsError with the reference's
    message unless override=True."""
    import json

    g = np.load(_util_golden() / "history20.npz")
    path = tmp_path / "HISTORY_singlemol_short"
    path.write_bytes(g["file_bytes"].tobytes())
    traj = DLPOLY(path)
    traj.analysis_output = {3: {"0": {"no_of_atoms": 168, "centre_of_mass": np.array([1.0, 2.0, 3.0]),
                                      "windows": {"diameters": None, "centre_of_mass": None}}}}
    out = tmp_path / "analysis"
    traj.save_analysis(out)
    written = tmp_path / "analysis.json"
    assert json.loads(written.read_text()) == {"3": {"0": {"no_of_atoms": 168, "centre_of_mass": [1.0, 2.0, 3.0],
                                                          "windows": {"diameters": None, "centre_of_mass": None}}}}
    meta = json.loads((_util_golden() / "history_analysis_3frames.meta.json").read_text())
    with pytest.raises(FileExistsError) as exc:
        traj.save_analysis(out)
    assert str(exc.value) == meta["second_save"].replace("<dir>", str(tmp_path))
    with pytest.raises(FileExistsError):
        traj.save_analysis(written)                     # the name already carries .json: not doubled
    traj.analysis_output[3]["0"]["no_of_atoms"] = 1
    traj.save_analysis(out, override=True)
    assert json.loads(written.read_text())["3"]["0"]["no_of_atoms"] == 1
    traj.save_analysis(tmp_path / "other.json.bak")     # '.json' in the name: kept as it is
    assert (tmp_path / "other.json.bak").is_file()


@pytest.mark.gpu
def test_saved_analysis_equals_the_reference_file(tmp_path):
    """DLPOLY.analysis + save_analysis on three frames of the reference's own trajectory: the JSON text
    is the one the reference writes (tests/golden/history_analysis_3frames.json, make_golden.py json) --
    same keys in the same order, every float with the same digits."""
    import json

    g = np.load(_util_golden() / "history20.npz")
    path = tmp_path / "HISTORY_singlemol_short"
    path.write_bytes(g["file_bytes"].tobytes())
    meta = json.loads((_util_golden() / "history_analysis_3frames.meta.json").read_text())
    traj = DLPOLY(path)
    traj.analysis(frames=meta["frames"], swap_atoms={"he": "H"}, forcefield="opls")
    traj.save_analysis(tmp_path / "mine")
    mine = (tmp_path / "mine.json").read_text()
    theirs = (_util_golden() / "history_analysis_3frames.json").read_text()
    assert json.loads(mine) == json.loads(theirs)
    assert mine == theirs
