#!/usr/bin/env python3
"""Kernel timeline of ONE device LM trial from a rocprofv3 --kernel-trace CSV of tools/lm_profile.py (developer tool):
   python tools/lm_trace.py <kernel_trace.csv>  -> name, duration, gap to the previous kernel (us)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one trial = from one lm_decide_kernel (exclusive) to the next (inclusive).  The device-steered loop queues one speculative trial behind
# the end of a solve, which drains as empty launches: take the trial of MEDIAN kernel time among those of the commonest launch count
idx = [i for i, r in enumerate(rows) if "lm_decide_kernel" in r["Kernel_Name"]]
if len(idx) < 3:
    sys.exit("fewer than three lm_decide_kernel launches in the trace")
segs = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1: b + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    segs.append((len(seg), busy, a, b))
segs = [g for g in segs if g[0] >= 5] or segs   # whole trials only (lm_profile.py also times lm_decide on its own); the fused trial has 6 launches
count = max(set(n for n, _, _, _ in segs), key=[n for n, _, _, _ in segs].count)
cand = sorted((busy, a, b) for n, busy, a, b in segs if n == count)
_, a, b = cand[len(cand) // 2]
seg = rows[a + 1: b + 1]
t_prev = int(rows[a]["End_Timestamp"])
busy = gaps = 0.0
lines = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcs::", "")[:60]
    d, g = (e - s) / 1e3, (s - t_prev) / 1e3
    busy += d
    gaps += max(g, 0.0)
    lines.append((name, d, max(g, 0.0)))
    t_prev = e
print(f"one trial (median of {len(cand)} with {count} launches): {len(seg)} launches, kernels {busy:.1f} us, gaps {gaps:.1f} us, span {busy + gaps:.1f} us")
for name, d, g in lines:   # in launch order
    print(f"  {name:60s} kernel {d:8.1f} us   gap before {g:6.1f} us")
