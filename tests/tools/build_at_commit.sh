#!/bin/bash
# The kernels of an earlier commit as a variant library, for same-box comparisons: build_at_commit.sh NAME COMMIT
# -> tests/tools/libpw_var_NAME.so (pw_kernels.hip of COMMIT with the other objects of the current build)
set -e
name=$1; commit=$2
root=$(cd "$(dirname "$0")/../.." && pwd)
tmp=/tmp/pw_at_$name; rm -rf $tmp; mkdir -p $tmp
cd $root
git archive $commit pywindow_amd/csrc include | tar -x -C $tmp
c=$root/pywindow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c $tmp/pywindow_amd/csrc/pw_kernels.hip -o /tmp/pwk_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread /tmp/pwk_$name.o $c/pw_kernels_big.o $c/pw_rebuild.o $c/pw_shape.o $c/pw_history.o $c/pw_hostpath.o -o $root/tests/tools/libpw_var_$name.so
echo built $root/tests/tools/libpw_var_$name.so from $commit
