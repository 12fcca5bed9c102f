#!/usr/bin/env python3
"""MFMA-busy per kernel from a rocprofv3 PMC pass:  python tools/pmc_mfma_busy.py <counter_collection.csv> [n_cus]
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x CUs x 4 SIMDs)  (the gfx94x derived-counter formula; ROCm 7.2 ships
no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots").  On this box GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs
(its per-launch value is ~8x the kernel's duration in shader cycles), so the column "per-XCD norm." divides by
(GRBM_GUI_ACTIVE / 8) instead; that figure agrees with the utilisation derived from flops / time / peak."""
import csv, re, sys
from collections import defaultdict

n_cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        key = (name, r.get("Grid_Size", ""))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[key] += 1
print(f"{'kernel':60s} {'grid':>9s} {'n':>5s} {'MFMA busy %':>11s} {'per-XCD norm. %':>16s} {'GUI_ACTIVE/launch':>18s}")
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:14]:
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if gui <= 0:
        continue
    raw = 100.0 * busy / (gui * n_cus * 4)
    print(f"{key[0]:60s} {key[1]:>9s} {cnt[key]:5d} {raw:11.2f} {raw * 8:16.2f} {gui / max(1, cnt[key]):18.0f}")
