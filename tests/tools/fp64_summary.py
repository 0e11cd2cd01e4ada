"""rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
SQ_INSTS_VALU_TRANS_F64 counter CSV -> <tag>_fp64_counters.json
   python tests/tools/fp64_summary.py counters.csv units out.json
What the vector ALUs really execute in double precision, per launch of the pipeline (kernels told apart by workgroup size
and stage mask, averaged over dispatches).  The counters count WAVE-level instructions; executed flop = (2 FMA + MUL + ADD
+ TRANS) x 64 lanes -- lanes switched off by the execution mask count too, so this is what the ALUs are occupied with, an
upper bound of the useful arithmetic.  (rocprofv3's own derived metric of the same counters: SQ_INSTS_VALU_FLOPS_FP64.)"""
import csv
import json
import re
import sys
from collections import defaultdict

path, units, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    if "pw_analyse_kernel" not in r["Kernel_Name"]:
        continue
    m = re.search(r"pw_analyse_kernel<(\d+), (\d+)u?>", r["Kernel_Name"])
    acc[(int(r["Workgroup_Size"]), int(m.group(2)) if m else -1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {37: "chains", 98: "average", 120: "windows", 122: "windows+average"}
kernels, total = {}, defaultdict(float)
for key, vals in sorted(acc.items()):
    name = names.get(key[1], "<%d, %d>" % (key[0] // 64, key[1]))
    # (a consumer dispatched ahead of its producer under the serialising profiler retires next to nothing: left out, as in
    # counter_summary.py)
    keep = list(range(len(next(iter(vals.values())))))
    if "SQ_INSTS_VALU" in vals and len(vals["SQ_INSTS_VALU"]) > 1:
        ref = sorted(vals["SQ_INSTS_VALU"])[len(vals["SQ_INSTS_VALU"]) // 2]
        keep = [i for i, x in enumerate(vals["SQ_INSTS_VALU"]) if x >= 0.1 * ref]
    row = {"dispatches": len(keep)}
    for c, v in vals.items():
        vv = [v[i] for i in keep if i < len(v)] or v
        row[c] = sum(vv) / len(vv)
    fma, mul, add, tr = (row.get("SQ_INSTS_VALU_" + k + "_F64", 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
    row["fp64_wave_instructions"] = fma + mul + add + tr
    row["executed_fp64_flop"] = (2.0 * fma + mul + add + tr) * 64.0
    row["fp64_share_of_valu"] = (fma + mul + add + tr) / row["SQ_INSTS_VALU"] if row.get("SQ_INSTS_VALU") else None
    kernels[name] = row
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64",
              "fp64_wave_instructions", "executed_fp64_flop"):
        total[c] += row.get(c, 0.0)
json.dump({"note": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 over tests/tools/run_stage.py 15 %d 4; "
                   "wave-level instruction counts per launch, averaged over dispatches; executed_fp64_flop = (2 FMA + MUL + ADD + "
                   "TRANS) x 64 lanes, inactive lanes included" % units,
           "kernels": kernels, "per_launch": dict(total), "units_per_launch": units,
           "executed_fp64_flop_per_unit": total["executed_fp64_flop"] / units}, open(out, "w"), indent=1)
print(json.dumps({k: {"executed_fp64_flop": v["executed_fp64_flop"], "fp64_share_of_valu": v["fp64_share_of_valu"]} for k, v in kernels.items()}, indent=1))
