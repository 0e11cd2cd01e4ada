#!/bin/bash
# Register / scratch usage of every device function of pw_kernels.hip (compiler remarks), one line each.
# usage: tests/tools/regs.sh [extra hipcc flags]
cd "$(dirname "$0")/../../pywindow_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c pw_kernels.hip -o /tmp/pw_regs.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import sys, re, subprocess
txt = sys.stdin.read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" [")[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"^void ", "", dem)
    dem = re.sub(r"\(anonymous namespace\)::", "", dem)
    dem = re.sub(r"\(.*", "", dem)[:100]
    g = lambda k: re.search(k + r": (\S+)", b).group(1)
    print("%-100s V%4s A%3s S%4s scratch%5s vspill%3s sspill%3s occ%2s" % (dem, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"),
          g(r"ScratchSize .bytes/lane."), g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy .waves/SIMD.")))
'
