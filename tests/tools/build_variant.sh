#!/bin/bash
# Build a variant of the library with extra compiler flags: tests/tools/build_variant.sh NAME -DFOO ...
# -> tests/tools/libpw_var_NAME.so (run with PW_LIB=<path> through tests/tools/with_lib.py)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pywindow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c $c/pw_kernels.hip -o /tmp/pwk_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread /tmp/pwk_$name.o $c/pw_kernels_big.o $c/pw_rebuild.o $c/pw_shape.o $c/pw_history.o $c/pw_hostpath.o -o $root/tests/tools/libpw_var_$name.so
echo built $root/tests/tools/libpw_var_$name.so
