"""pywindow_amd -- MI355X-native engine for pywindow's ``full_analysis()`` hot path.

The package keeps the reference's ``Molecule`` / ``MolecularSystem`` / ``DLPOLY``
API surface for that path and executes it with hand-written FP64 HIP kernels for
gfx950 (``csrc/``), through a C ABI (``include/pywindow_amd.h``) bound with
ctypes.  Nothing here falls back to a CPU: without the HIP library or a device every call raises.  (The one
CPU path that exists is explicit -- a context created with ``device=-1`` runs the same kernel source compiled for
the host, ``csrc/pw_hostpath.cpp``; nothing selects it but the caller.)
"""

import os as _os

# several kernels of one analysis run side by side on separate HIP streams; the runtime must be
# told before it initialises (harmless if the host application already set it)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

from .molecular import MolecularSystem, Molecule  # noqa: E402
from .trajectory import DLPOLY  # noqa: E402
from .utilities import (  # noqa: E402
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
