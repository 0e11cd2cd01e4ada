"""``Molecule`` / ``MolecularSystem`` with the reference's API surface for the hot
path (reference molecular.py:60-352, 554-836), backed by the HIP engine.

Only what ``full_analysis`` and the trajectory driver need is here: loading a
system dict, force-field key handling, ``system_to_molecule`` and the nine
``calculate_*`` methods with the reference's ``properties`` schema.  File
writers and the periodic ``rebuild_system`` are outside the accelerated path.
"""

from __future__ import annotations

from copy import deepcopy

import numpy as np

from . import _lib, engine
from .element_data import OPLS_KEY_TO_ELEMENT


class _AtomKeyError(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class _AtomKeyConflictError(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


class _ForceFieldError(Exception):
    def __init__(self, message: str) -> None:
        self.message = message


def _is_number(s: str) -> bool:
    try:
        float(s)
    except ValueError:
        return False
    return True


def decipher_atom_key(atom_key: str, forcefield: str) -> str:
    """Force-field atom key -> element (reference utilities.py:267-341)."""
    ff = forcefield.upper()
    if ff in ("DLF", "DL_F"):
        head = ""
        for ch in atom_key:
            if head and _is_number(ch):
                break
            head += ch
        else:
            # the reference indexes past the end here; a key without digits is an error there too
            raise IndexError("string index out of range")
        return "".join(c for c in head if not _is_number(c) and c != "?")
    if ff in ("OPLS", "OPLSAA", "OPLS2005", "OPLS3"):
        if atom_key in ("ne", "he", "na"):
            raise _AtomKeyConflictError(
                f"One of the OPLS conflicting atom_keys has occured '{atom_key}'. "
                "For how to solve this issue see the manual or "
                "MolecularSystem._atom_key_swap() doc string."
            )
        try:
            return OPLS_KEY_TO_ELEMENT[atom_key]
        except KeyError:
            raise _AtomKeyError(
                f"OPLS atom key {atom_key} was not found in OPLS keys dictionary."
            ) from None
    raise _ForceFieldError(
        f"Unfortunetely, '{forcefield}' forcefield is not supported by pyWINDOW."
    )


class Molecule:
    """A single discrete molecule; every ``calculate_*`` runs on the GPU."""

    def __init__(self, mol: dict, system_name: str, mol_id) -> None:
        self.mol = mol
        self.no_of_atoms = len(mol["elements"])
        self.elements = mol["elements"]
        if "atom_ids" in mol:
            self.atom_ids = mol["atom_ids"]
        self.coordinates = mol["coordinates"]
        self.parent_system = system_name
        self.molecule_id = mol_id
        self.properties = {"no_of_atoms": self.no_of_atoms}
        self._cache: dict[int, np.void] = {}

    # one launch per stage set; records are cached so that chained calls
    # (pore volume -> pore diameter, ...) do not recompute -- the reference does
    # recompute (molecular.py:279, 317) but is deterministic, so results agree
    def _record(self, stages: int):
        for have, rec in self._cache.items():
            if have & stages == stages:
                return rec
        extra: list = []
        rec = engine.analyse([(self.elements, self.coordinates)], stages, extra=extra)[0]
        engine.raise_on_capacity(rec)
        self._cache = {stages: rec}
        self._more = engine.extra_by_unit(extra).get(0)      # windows beyond what a record holds
        return rec

    def _invalidate(self):
        self._cache = {}

    def full_analysis(self, ncpus: int = 1) -> dict:  # noqa: ARG002
        """All nine properties in ONE kernel launch (reference molecular.py:156-202)."""
        rec = self._record(_lib.STAGE_ALL)
        if int(rec["status"]) & _lib.ST_NEGATIVE_PORE:
            # the reference fills the dict in order and SciPy stops it at pore_diameter_opt
            # (molecular.py:196-199 -> utilities.py:422): same entries, same exception
            self.calculate_centre_of_mass()
            self.calculate_maximum_diameter()
            self.calculate_average_diameter()
            self.calculate_pore_diameter()
            self.calculate_pore_volume()
            engine.raise_like_reference(rec)
        self._fill_from(rec)
        if int(rec["status"]) & _lib.ST_TOO_FEW_POINTS:
            # molecular.py:200 -> utilities.py:1428-1431: KDTree.query(k=10) on fewer than ten sampling vectors
            del self.properties["windows"]
            raise ValueError("k must be less than or equal to the number of training points")
        return self.properties

    def _fill_from(self, rec):
        engine.warn_like_reference(rec)
        props = engine.record_to_properties(rec, more=getattr(self, "_more", None))
        self.MW = float(rec["mw"])
        self.centre_of_mass = props["centre_of_mass"]
        md = props["maximum_diameter"]
        self.maxd_atom_1, self.maxd_atom_2, self.maximum_diameter = md["atom_1"], md["atom_2"], md["diameter"]
        self.average_diameter = props["average_diameter"]
        self.pore_diameter, self.pore_closest_atom = props["pore_diameter"]["diameter"], props["pore_diameter"]["atom"]
        self.pore_volume = props["pore_volume"]
        po = props["pore_diameter_opt"]
        self.pore_diameter_opt, self.pore_opt_closest_atom, self.pore_opt_COM = (
            po["diameter"], po["atom_1"], po["centre_of_mass"])
        self.pore_volume_opt = props["pore_volume_opt"]
        # key order of the reference's dict
        for key in ("centre_of_mass", "maximum_diameter", "average_diameter", "pore_diameter",
                    "pore_volume", "pore_diameter_opt", "pore_volume_opt", "windows"):
            self.properties[key] = props[key]

    def molecular_weight(self) -> float:
        self.MW = float(self._record(_lib.STAGE_BASIC)["mw"])
        return self.MW

    def calculate_centre_of_mass(self) -> np.ndarray:
        self.centre_of_mass = np.array(self._record(_lib.STAGE_BASIC)["com"])
        self.properties["centre_of_mass"] = self.centre_of_mass
        return self.centre_of_mass

    def calculate_maximum_diameter(self) -> float:
        r = self._record(_lib.STAGE_BASIC)
        self.maxd_atom_1, self.maxd_atom_2, self.maximum_diameter = int(r["maxd_i"]), int(r["maxd_j"]), float(r["maxd"])
        self.properties["maximum_diameter"] = {
            "diameter": self.maximum_diameter, "atom_1": self.maxd_atom_1, "atom_2": self.maxd_atom_2}
        return self.maximum_diameter

    def calculate_average_diameter(self) -> float:
        self.average_diameter = float(self._record(_lib.STAGE_AVG)["avg_d"])
        self.properties["average_diameter"] = self.average_diameter
        return self.average_diameter

    def calculate_pore_diameter(self) -> float:
        r = self._record(_lib.STAGE_BASIC)
        self.pore_diameter, self.pore_closest_atom = float(r["pore_d"]), int(r["pore_atom"])
        self.properties["pore_diameter"] = {"diameter": self.pore_diameter, "atom": self.pore_closest_atom}
        return self.pore_diameter

    def calculate_pore_volume(self) -> float:
        self.calculate_pore_diameter()
        self.pore_volume = float(self._record(_lib.STAGE_BASIC)["pore_vol"])
        self.properties["pore_volume"] = self.pore_volume
        return self.pore_volume

    def calculate_pore_diameter_opt(self) -> float:
        r = self._record(_lib.STAGE_OPT)
        engine.raise_like_reference(r)
        self.pore_diameter_opt = float(r["pore_opt_d"])
        self.pore_opt_closest_atom = int(r["pore_opt_atom"])
        self.pore_opt_COM = np.array(r["pore_opt_c"])
        self.properties["pore_diameter_opt"] = {
            "diameter": self.pore_diameter_opt, "atom_1": self.pore_opt_closest_atom,
            "centre_of_mass": self.pore_opt_COM}
        return self.pore_diameter_opt

    def calculate_pore_volume_opt(self) -> float:
        self.calculate_pore_diameter_opt()
        self.pore_volume_opt = float(self._record(_lib.STAGE_OPT)["pore_vol_opt"])
        self.properties["pore_volume_opt"] = self.pore_volume_opt
        return self.pore_volume_opt

    def calculate_windows(self, ncpus: int = 1):  # noqa: ARG002
        r = self._record(_lib.STAGE_WINDOWS)
        engine.raise_like_reference(r)
        engine.warn_like_reference(r)
        if int(r["status"]) & _lib.ST_TOO_FEW_POINTS:
            raise ValueError("k must be less than or equal to the number of training points")
        win = engine.windows_of(r, getattr(self, "_more", None))
        if win is not None:
            self.properties["windows"] = {"diameters": win[0], "centre_of_mass": win[1]}
            return win[0]
        self.properties["windows"] = {"diameters": None, "centre_of_mass": None}
        return None

    def _align_to_principal_axes(self, align_molsys: bool = False) -> None:
        """Reference molecular.py:204-213.  There the result -- a ``(coordinates, rotations)`` tuple -- is
        assigned to ``self.coordinates[0]``, which numpy refuses; here the molecule takes the aligned
        coordinates (what the method's name promises) and keeps the rotations."""
        if align_molsys:
            raise NotImplementedError
        from .utilities import align_principal_ax

        aligned, self.principal_axes_rotations = align_principal_ax(self.elements, self.coordinates)
        self.coordinates = aligned
        self.mol["coordinates"] = aligned
        self.aligned_to_principal_axes = True
        self._invalidate()

    def shift_to_origin(self) -> None:
        com = self.calculate_centre_of_mass()
        self.coordinates = np.asarray(self.coordinates, float) - np.array([com] * self.no_of_atoms)
        self.mol["coordinates"] = self.coordinates
        self._invalidate()


class MolecularSystem:
    """Container of a (possibly multi-molecule) system (reference molecular.py:554-836)."""

    def __init__(self) -> None:
        self.system_id = 0
        self.system: dict = {}

    @classmethod
    def load_system(cls, dict_: dict, system_id: str | int = "system") -> "MolecularSystem":
        obj = cls()
        obj.system = dict_
        obj.system_id = system_id
        return obj

    @classmethod
    def load_file(cls, filepath) -> "MolecularSystem":
        """Load an ``.xyz`` or ``.pdb`` file (reference molecular.py:596-623 with the readers of
        io_tools.py:106-182; CRYST1 becomes ``unit_cell`` and ``lattice``)."""
        import pathlib

        from .rebuild import unit_cell_to_lattice_array

        path = pathlib.Path(filepath)
        lines = path.read_text().splitlines(keepends=True)
        system: dict = {}
        if path.suffix == ".xyz":
            body = [ln.split() for ln in lines[2:]]
            try:
                system["elements"] = np.array([b[0] for b in body])
                system["coordinates"] = np.array([[float(b[1]), float(b[2]), float(b[3])] for b in body])
            except IndexError:
                raise ValueError("The XYZ file is corrupted in some way (empty line at the end, or a trajectory).") from None
        elif path.suffix == ".pdb":
            if sum(ln.count("END ") for ln in lines) > 1:
                raise ValueError("Multiple 'END' statements were found in this PDB file.")
            atoms = [ln for ln in lines if ln[:6] in ("HETATM", "ATOM  ")]
            system["remarks"] = [ln for ln in lines if ln[:6] == "REMARK"]
            system["unit_cell"] = np.array([
                float(x) for ln in lines if ln[:6] == "CRYST1"
                for x in (ln[6:15], ln[15:24], ln[24:33], ln[33:40], ln[40:47], ln[47:54])
            ])
            if system["unit_cell"].any():
                system["lattice"] = unit_cell_to_lattice_array(system["unit_cell"])
            system["atom_ids"] = np.array([ln[12:16].strip() for ln in atoms], dtype="<U8")
            system["elements"] = np.array([ln[76:78].strip() for ln in atoms], dtype="<U8")
            system["coordinates"] = np.array([[float(ln[30:38]), float(ln[38:46]), float(ln[46:54])] for ln in atoms])
        else:
            raise ValueError(f"unsupported file type {path.suffix!r} (xyz and pdb are read natively)")
        obj = cls()
        obj.system = system
        obj.filename = path.name
        obj.system_id = path.name.split(".")[0]
        return obj

    def rebuild_system(self, override: bool = False) -> "MolecularSystem":
        """Re-assemble the molecules of a periodic system through the cell faces
        (reference molecular.py:672-708); the 3x3x3 supercell and the bonded-neighbour walk
        run on the GPU (csrc/pw_rebuild.hpp)."""
        from .rebuild import discrete_molecules

        discrete = discrete_molecules(self.system, rebuild=True)
        coordinates = np.array([], dtype=np.float64).reshape(0, 3)
        atom_ids = np.array([])
        elements = np.array([])
        for mol in discrete:
            coordinates = np.concatenate([coordinates, mol["coordinates"]], axis=0)
            atom_ids = np.concatenate([atom_ids, mol["atom_ids"]], axis=0)
            elements = np.concatenate([elements, mol["elements"]], axis=0)
        rebuilt = {"coordinates": coordinates, "atom_ids": atom_ids, "elements": elements}
        if override is True:
            self.system.update(rebuilt)
        return self.load_system(rebuilt)

    def make_modular(self, rebuild: bool = False) -> None:
        """Populate ``self.molecules`` with the discrete molecules of the system
        (reference molecular.py:798-824)."""
        from .rebuild import discrete_molecules

        dis = discrete_molecules(self.system, rebuild=True if rebuild is True else None)
        self.no_of_discrete_molecules = len(dis)
        self.molecules = {}
        for i in range(len(dis)):
            self.molecules[i] = Molecule(dis[i], str(self.system_id), i)

    def swap_atom_keys(self, swap_dict: dict, dict_key: str = "atom_ids") -> None:
        """Reference molecular.py:710-749."""
        if "atom_ids" not in self.system:
            dict_key = "elements"
        for atom_key in range(len(self.system[dict_key])):
            for key in swap_dict:
                if self.system[dict_key][atom_key] == key:
                    self.system[dict_key][atom_key] = swap_dict[key]

    def decipher_atom_keys(self, forcefield: str = "DLF", dict_key: str = "atom_ids") -> None:
        """Reference molecular.py:751-796."""
        if "atom_ids" not in self.system:
            dict_key = "elements"
        temp = deepcopy(self.system[dict_key])
        for element in range(len(temp)):
            temp[element] = str(decipher_atom_key(temp[element], forcefield=forcefield))
        self.system["elements"] = temp

    def system_to_molecule(self) -> Molecule:
        return Molecule(self.system, self.system_id, 0)
