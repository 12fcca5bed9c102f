#!/usr/bin/env python3
"""Averages a rocprofv3 counter_collection.csv per kernel: `python tools/pmc_summary.py <csv> <COUNTER>`."""
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != sys.argv[2]:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:50]
        key = (name, r.get("Grid_Size", ""))
        a = acc[key]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print(f"counter {sys.argv[2]}: per-kernel average per dispatch")
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{k[0]:50s} grid={k[1]:>8s} n={n:6d} avg={v/n:14.1f} total={v:16.1f}")
