"""Summarise a rocprofv3 kernel-trace CSV as a per-kernel timeline (ms from the first start)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    name = r["Kernel_Name"][:48]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:9.3f} -> {e:9.3f}  ({e - s:7.3f} ms)  grid {r.get('Grid_Size_X', '?'):>7} wg {r.get('Workgroup_Size_X', '?'):>4}  {name}")
