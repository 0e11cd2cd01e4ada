"""Build the CPU-only test harnesses (g++, no GPU needed): the kernel headers of
pywindow_amd/csrc compiled with a one-thread team.  Test infrastructure."""
import pathlib
import subprocess

HERE = pathlib.Path(__file__).resolve().parent
FLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared"]
TARGETS = {
    "libblasprobe.so": "blas_probe.cpp",
    "liblbprobe.so": "lbfgsb_probe.cpp",
    "libmathprobe.so": "math_probe.cpp",
    "libunitprobe.so": "unit_probe.cpp",
    "librebuildprobe.so": "rebuild_probe.cpp",
    "libshapeprobe.so": "shape_probe.cpp",
    # the unit pipeline WITHOUT the scans' candidate lists and block tests (tests/test_scan_lists.py: the same records)
    "libunitprobe_nolists.so": ("unit_probe.cpp", "-DPW_NO_SCAN_LISTS"),
}


def build(force=False):
    csrc = HERE.parent.parent / "pywindow_amd" / "csrc"
    newest = max(p.stat().st_mtime for p in list(csrc.glob("*.hpp")) + list(HERE.glob("*.cpp")))
    for so, src in TARGETS.items():
        extra = []
        if isinstance(src, tuple):
            src, *extra = src
        out = HERE / so
        if force or not out.exists() or out.stat().st_mtime < newest:
            subprocess.check_call(["g++", *FLAGS, *extra, "-o", str(out), str(HERE / src)])


if __name__ == "__main__":
    build()
