#!/bin/bash
# Per-function resources (VGPRs, scratch, static scratch stores / loads, calls) of pw_kernels.hip's gfx950 code, kernels
# and out-of-line device functions alike.   usage: tests/tools/isa.sh [extra hipcc flags]   (listing: /tmp/isa/*.s)
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/isa && cd /tmp/isa || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -save-temps=obj "$@" \
    -c $root/pywindow_amd/csrc/pw_kernels.hip -o /tmp/isa/pw_kernels.o 2>&1 | grep -v warning | head -20
python3 $root/tests/tools/isa_functions.py /tmp/isa/pw_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
