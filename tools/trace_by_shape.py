#!/usr/bin/env python3
"""Groups a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup) and prints avg/min durations; deletes nothing."""
import csv, re, sys
from collections import defaultdict
rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:46]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    v.sort()
    print(f"{k[0]:46s} grid={k[1]:>8s} wg={k[2]:>4s} n={len(v):6d} tot={sum(v)/1e6:8.2f}ms {100*sum(v)/tot:5.1f}% avg={sum(v)/len(v)/1e3:7.2f}us p10={v[len(v)//10]/1e3:7.2f} min={v[0]/1e3:7.2f}")
