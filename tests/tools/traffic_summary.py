"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs -> profiles/<tag>_hbm_traffic.json.
   python tests/tools/traffic_summary.py fetch.csv write.csv units out.json
Counter values are KB per dispatch; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950.  Dispatches are grouped by (grid, workgroup) = the three launches of the pipeline."""
import csv
import json
import sys
from collections import defaultdict

fetch, write, units, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
roles = {64: "A optimiser chains (1 wave/unit)"}
acc = defaultdict(lambda: defaultdict(list))
for path in (fetch, write):
    for r in csv.DictReader(open(path)):
        if "pw_analyse_kernel" not in r["Kernel_Name"]:
            continue
        # the launches are told apart by the template arguments in the kernel name (<waves, stage mask>)
        m = __import__("re").search(r"pw_analyse_kernel<(\d+), (\d+)u?>", r["Kernel_Name"])
        key = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(m.group(2)) if m else -1)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels = {}
total = 0.0
for key, vals in sorted(acc.items()):
    if key[1] == 64:
        name = roles[64]
    elif key[2] == 120:
        name = "C window search (4 waves/unit, persistent consumer)"
    elif key[2] == 122:
        name = "C window search + average diameter (4 waves/unit, persistent consumer)"
    elif key[2] == 98:
        name = "B average diameter (4 waves/unit)"
    else:
        name = f"pw_analyse_kernel<{key[1] // 64}, {key[2]}>"
    f = 2.0 * sum(vals["FETCH_SIZE"]) / max(len(vals["FETCH_SIZE"]), 1)
    w = sum(vals["WRITE_SIZE"]) / max(len(vals["WRITE_SIZE"]), 1)
    kernels[name] = {"grid": key[0], "workgroup": key[1], "FETCH_SIZE_KB_x2": f, "WRITE_SIZE_KB": w,
                     "dispatches": len(vals["FETCH_SIZE"])}
    total += (f + w) * 1024.0
json.dump({
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tests/tools/run_stage.py; KB per "
            "dispatch averaged over dispatches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950). The counters "
            "include Infinity-Cache hits: k-NN rows, per-team workspaces and register save areas are re-used "
            "by every unit a team processes and stay on-die.",
    "kernels": kernels, "per_launch_bytes": total, "units_per_launch": units}, open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1), total)
