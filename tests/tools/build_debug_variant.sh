#!/bin/bash
# The library with line tables in its device code (for rocgdb): tests/tools/libpw_var_dbg.so
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
c=$root/pywindow_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -gline-tables-only"
/opt/rocm/bin/hipcc $F "$@" -c $c/pw_kernels.hip -o /tmp/pwk_dbg.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread /tmp/pwk_dbg.o $c/pw_kernels_big.o $c/pw_rebuild.o $c/pw_shape.o $c/pw_history.o $c/pw_hostpath.o -o $root/tests/tools/libpw_var_dbg.so
echo built $root/tests/tools/libpw_var_dbg.so
