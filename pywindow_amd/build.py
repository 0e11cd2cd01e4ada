"""Build libpywindow_hip.so in-tree with hipcc for gfx950 (no CPU fallback exists).

    python -m pywindow_amd.build            # rebuild if sources are newer
"""

from __future__ import annotations

import os
import pathlib
import shutil
import subprocess
import sys

PKG = pathlib.Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "libpywindow_hip.so"
SOURCES = ["pw_kernels.hip", "pw_kernels_big.hip", "pw_rebuild.hip", "pw_shape.hip", "pw_history.cpp", "pw_hostpath.cpp"]
# -ffp-contract=off: the numerical core relies on explicit fma() only (pw_common.hpp)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and pathlib.Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def stale() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "pywindow_amd.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> pathlib.Path:
    if not force and not stale():
        return OUT
    # one translation unit per source, compiled separately (mixing "-x hip" and "-x c++" in one
    # hipcc command silently drops --offload-arch) and linked by hipcc
    obj_k = CSRC / "pw_kernels.o"
    obj_b = CSRC / "pw_kernels_big.o"
    obj_r = CSRC / "pw_rebuild.o"
    obj_s = CSRC / "pw_shape.o"
    obj_h = CSRC / "pw_history.o"
    obj_c = CSRC / "pw_hostpath.o"
    hip_flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c"]
    cmds = [
        [hipcc(), *hip_flags, str(CSRC / "pw_kernels.hip"), "-o", str(obj_k)],
        [hipcc(), *hip_flags, str(CSRC / "pw_kernels_big.hip"), "-o", str(obj_b)],
        [hipcc(), *hip_flags, str(CSRC / "pw_rebuild.hip"), "-o", str(obj_r)],
        [hipcc(), *hip_flags, str(CSRC / "pw_shape.hip"), "-o", str(obj_s)],
        ["g++", "-O2", "-std=c++17", "-fPIC", "-c", str(CSRC / "pw_history.cpp"), "-o", str(obj_h)],
        # the explicit host path (pw_context_create(-1)): the unit pipeline for a one-lane team, g++
        ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-pthread", "-c", str(CSRC / "pw_hostpath.cpp"),
         "-o", str(obj_c)],
        [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", str(obj_k), str(obj_b), str(obj_r), str(obj_s), str(obj_h),
         str(obj_c), "-o", str(OUT)],
    ]
    # the translation units are independent: compile them side by side (PW_BUILD_JOBS, default 4), then link
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = max(1, int(os.environ.get("PW_BUILD_JOBS", "4")))
    with ThreadPoolExecutor(jobs) as pool:
        list(pool.map(run, cmds[:-1]))
    run(cmds[-1])
    text = subprocess.run(["strings", str(OUT)], capture_output=True, text=True).stdout
    if "amdgcn-amd-amdhsa--gfx950" not in text:
        raise RuntimeError("built library does not contain a gfx950 code object")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
