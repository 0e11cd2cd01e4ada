"""pywindow_amd -- MI355X-native engine for pywindow's ``full_analysis()`` hot path.

The package keeps the reference's ``Molecule`` / ``MolecularSystem`` / ``DLPOLY``
API surface for that path and executes it with hand-written FP64 HIP kernels for
gfx950 (``csrc/``), through a C ABI (``include/pywindow_amd.h``) bound with
ctypes.  There is no CPU implementation in this package.
"""

from .molecular import MolecularSystem, Molecule
from .trajectory import DLPOLY
from .utilities import (
    center_of_mass,
    find_average_diameter,
    find_windows,
    max_dim,
    molecular_weight,
    opt_pore_diameter,
    pore_diameter,
    shift_com,
    sphere_volume,
)

__all__ = [
    "DLPOLY",
    "MolecularSystem",
    "Molecule",
    "center_of_mass",
    "find_average_diameter",
    "find_windows",
    "max_dim",
    "molecular_weight",
    "opt_pore_diameter",
    "pore_diameter",
    "shift_com",
    "sphere_volume",
]
