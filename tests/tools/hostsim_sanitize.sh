#!/bin/bash
# The kernel headers compiled for the host (tests/hostsim) under AddressSanitizer + UBSan (GPU
# sanitizers are not available on the pool): builds instrumented probes, runs the hostsim tests
# against them, restores the plain probes.  Run from the repo root.
set -e
out=/tmp/asan_hostsim; mkdir -p $out
flags="-O1 -g -std=c++17 -ffp-contract=off -mfma -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer"
for src in unit_probe rebuild_probe shape_probe lbfgsb_probe math_probe blas_probe; do
  case $src in unit_probe) so=libunitprobe.so;; rebuild_probe) so=librebuildprobe.so;; shape_probe) so=libshapeprobe.so;;
    lbfgsb_probe) so=liblbprobe.so;; math_probe) so=libmathprobe.so;; blas_probe) so=libblasprobe.so;; esac
  g++ $flags -o $out/$so tests/hostsim/$src.cpp
done
cp $out/*.so tests/hostsim/ && touch tests/hostsim/*.so
asan=$(gcc -print-file-name=libasan.so); ubsan=$(gcc -print-file-name=libubsan.so)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD="$asan $ubsan" python -m pytest -x -q -s -m "not gpu" -p no:cacheprovider \
  tests/test_hostsim_golden.py tests/test_options.py tests/test_rebuild.py tests/test_shape.py tests/test_math.py \
  tests/test_lbfgsb_lockstep.py 2>&1 | grep -i "runtime error\|AddressSanitizer\|passed\|failed" | sort | uniq -c
python -c "
import importlib.util
s = importlib.util.spec_from_file_location('hb', 'tests/hostsim/build.py'); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); m.build(force=True)"
