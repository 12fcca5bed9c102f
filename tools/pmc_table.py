#!/usr/bin/env python3
"""Per-kernel table of every counter in a rocprofv3 counter_collection.csv (average per dispatch):
    python tools/pmc_table.py <counter_collection.csv> [substring of kernel names to keep]"""
import csv, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
keep = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if keep and keep not in name:
            continue
        key = (name[:70], r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[key][r["Counter_Name"]] += 1
counters = sorted({c for v in acc.values() for c in v})
print("kernel | grid | n | " + " | ".join(counters))
for key in sorted(acc, key=lambda k: -sum(acc[k].values())):
    n = max(cnt[key].values())
    print(f"{key[0]} | {key[1]} | {n} | " + " | ".join(f"{acc[key][c] / max(1, cnt[key][c]):.4g}" for c in counters))
