"""rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU counter CSV -> <tag>_instruction_counters.json
   python tests/tools/counter_summary.py counters.csv units out.json
Wave-level instruction counts per launch, averaged over the dispatches of each of the three launches of
the pipeline (told apart by grid and workgroup size)."""
import csv
import json
import sys
from collections import defaultdict

path, units, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    if "pw_analyse_kernel" not in r["Kernel_Name"]:
        continue
    m = __import__("re").search(r"pw_analyse_kernel<(\d+), (\d+)u?>", r["Kernel_Name"])
    acc[(int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(m.group(2)) if m else -1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels, total = {}, defaultdict(float)
for key, vals in sorted(acc.items()):
    name = ("A optimiser chains (1 wave/unit)" if key[1] == 64 else
            "C window search" if key[2] == 120 else "C window search + average diameter" if key[2] == 122 else
            "B average diameter" if key[2] == 98 else f"<{key[1] // 64}, {key[2]}>")
    row = {"grid": key[0], "workgroup": key[1]}
    # A counter pass serialises the kernels of the process: a consumer launch dispatched ahead of its producer then finds
    # no units, waits out its time limit and retires a few thousand instructions (seen once in round 4, once in round 5:
    # DESIGN.md section 7).  Such a dispatch -- under a tenth of the median of its kind in SQ_INSTS_VALU -- says nothing
    # about the kernel and is left out, and reported.
    keep = list(range(len(next(iter(vals.values())))))
    if "SQ_INSTS_VALU" in vals and len(vals["SQ_INSTS_VALU"]) > 1:
        ref = sorted(vals["SQ_INSTS_VALU"])[len(vals["SQ_INSTS_VALU"]) // 2]
        keep = [i for i, x in enumerate(vals["SQ_INSTS_VALU"]) if x >= 0.1 * ref]
        if len(keep) < len(vals["SQ_INSTS_VALU"]):
            row["dispatches_left_out"] = len(vals["SQ_INSTS_VALU"]) - len(keep)
    for c, v in vals.items():
        vv = [v[i] for i in keep if i < len(v)] or v
        row[c] = sum(vv) / len(vv)
        if c != "SQ_WAVES":
            total[c] += row[c]
    kernels[name] = row
json.dump({"note": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU over tests/tools/run_stage.py 15 %d 4; "
                   "wave-level instruction counts per launch, averaged over dispatches" % units,
           "kernels": kernels, "per_launch": dict(total), "units_per_launch": units,
           "valu_issue": {"simd_cycles_per_wave_instruction": 4, "simds": 1024, "clock_ghz": 2.4}}, open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
