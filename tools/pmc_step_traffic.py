#!/usr/bin/env python3
"""HBM bytes per decode step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-graph ...`.

    python tools/pmc_step_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> '<config json>' > profiles/rNN_pmc_step_traffic.json

Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KiB) counts HALF of the bytes of wide coalesced reads on gfx950 -> x2;
WRITE_SIZE (KiB) is taken as is.  Decode-step kernels = every dispatch between two `sampler_finish_kernel`s that belongs to
the step (projections, decode attention, sampler, embedding); steps = number of sampler_finish dispatches."""
import csv, json, re, sys
from collections import defaultdict

STEP_KERNELS = ("skinny_mfma_kernel", "dec_self_attn_kernel", "dec_cross_attn_kernel", "sampler_part_kernel",
                "sampler_finish_kernel", "embed_kernel")


def collect(path, counter):
    per = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            k = next((s for s in STEP_KERNELS if s in name), None)
            if k is None:
                continue
            if k == "skinny_mfma_kernel":
                k += " grid=" + r.get("Grid_Size", "?")
            per[k][0] += 1
            per[k][1] += float(r["Counter_Value"])
    return per


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
steps = fetch["sampler_finish_kernel"][0]
rows = {}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fb = fetch[k][1] * 1024 * 2 / max(1, steps)
    wb = write[k][1] * 1024 / max(1, steps)
    rows[k] = {"launches_per_step": round(fetch[k][0] / max(1, steps), 2), "fetch_bytes_per_step": round(fb), "write_bytes_per_step": round(wb)}
    tot += fb + wb
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thewhisper_amd.build import source_digest  # noqa: E402

print(json.dumps({"config": json.loads(sys.argv[3]), "kernel_source_sha256": source_digest(), "decode_steps": steps, "hbm_bytes_per_step": round(tot),
                  "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE KiB x 1024",
                  "per_kernel": rows}, indent=1))
