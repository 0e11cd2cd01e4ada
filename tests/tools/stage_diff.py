"""Print the in-kernel stage timers of a quick round next to the tracked ones.  usage: stage_diff.py gpurun_out/TAG/stages.json"""
import json, sys, pathlib
root = pathlib.Path(__file__).resolve().parents[2]
a = json.load(open(root / "profiles" / "r02_stage_timers.json"))["us_per_unit"]
t = open(sys.argv[1]).read()
b = json.loads(t[t.index("{"):])
print("kernel_ms", b["kernel_ms"])
for k, v in b["us_per_unit"].items():
    if abs(a.get(k, 0) - v) > 1.5:
        print("%-32s %8.2f -> %8.2f" % (k, a.get(k, 0), v))
