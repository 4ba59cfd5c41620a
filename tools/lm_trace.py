#!/usr/bin/env python3
"""Kernel timeline of ONE device LM trial from a rocprofv3 --kernel-trace CSV of tools/lm_profile.py (developer tool):
   python tools/lm_trace.py <kernel_trace.csv>  -> name, duration, gap to the previous kernel (us)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last complete trial: from the last-but-one lm_decide_kernel (exclusive) to the last one (inclusive)
idx = [i for i, r in enumerate(rows) if "lm_decide_kernel" in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("no two lm_decide_kernel launches in the trace")
seg = rows[idx[-2] + 1: idx[-1] + 1]
t_prev = int(rows[idx[-2]]["End_Timestamp"])
busy = gaps = 0.0
short = {}
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcs::", "")[:60]
    d, g = (e - s) / 1e3, (s - t_prev) / 1e3
    busy += d
    gaps += max(g, 0.0)
    key = name
    short.setdefault(key, [0, 0.0, 0.0])
    short[key][0] += 1; short[key][1] += d; short[key][2] += max(g, 0.0)
    t_prev = e
print(f"one trial: {len(seg)} launches, kernels {busy:.1f} us, gaps {gaps:.1f} us, span {busy + gaps:.1f} us")
for k, (n, d, g) in sorted(short.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:60s} x{n:3d}  kernels {d:8.1f} us   gaps before {g:7.1f} us")
