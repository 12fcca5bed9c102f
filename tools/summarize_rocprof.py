#!/usr/bin/env python3
"""Turns `rocprofv3 --kernel-trace --stats --output-format csv` output into the short markdown summaries kept in profiles/.

    python tools/summarize_rocprof.py gpurun_out/prof2/r01_kernel_stats.csv [bench.json-line file] > profiles/r01_....md
"""
import csv
import json
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*?)E+v", name)
    if m:
        t = m.group(2).replace("DF16b", "bf16,").replace("Li", "").replace("ELb0", ",false").replace("ELb1", ",true").replace("E", ",")
        t = t.replace("f,", "f32,") if t.startswith("f") else t
        return f"{m.group(1)}<{t.strip(',')}>"
    return name[:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | total ms | % | avg µs | min µs | max µs |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:22]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {100*float(r['TotalDurationNs'])/tot:.1f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    print(f"\nTotal kernel time in trace: {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} dispatches "
          "(includes weight generation/upload kernels of the bench set-up).")
    if len(sys.argv) > 2:
        line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
        j = json.loads(line)
        print("\nbench line of the profiled run:\n\n```json\n" + json.dumps(j, indent=1) + "\n```")


if __name__ == "__main__":
    main()
