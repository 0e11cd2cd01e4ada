#!/bin/bash
# A variant of the library whose SAMPLING translation unit is built with extra flags:
#   tests/tools/build_sampling_variant.sh NAME -DFOO ...  -> tests/tools/libpw_var_NAME.so   (run with PW_LIB=<path>)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pywindow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c $c/pw_kernels_sampling.hip -o /tmp/pws_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread $c/pw_kernels.o /tmp/pws_$name.o $c/pw_kernels_big.o $c/pw_rebuild.o $c/pw_shape.o $c/pw_history.o $c/pw_hostpath.o -o $root/tests/tools/libpw_var_$name.so
echo built $root/tests/tools/libpw_var_$name.so
