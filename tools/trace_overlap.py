#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: for one graph replay (the last complete one), how much of the step is
covered by >= 1 kernel, how much by >= 2, and which kernels run alone (the serial part of the schedule)."""
import csv, sys, glob, os
from collections import defaultdict
f = (glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv")) + glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mspi::", "")[:48]) for r in csv.DictReader(open(f))]
rows.sort()
# split into bursts separated by gaps > 200 us; take the median-length burst among the last 10 as "a step"
bursts, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if r[0] - end > 200000:
        bursts.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
bursts.append(cur)
cand = [b for b in bursts[-12:-1] if len(b) > 300]
b = sorted(cand, key=lambda x: x[-1][1] - x[0][0])[len(cand) // 2]
t0, t1 = b[0][0], max(r[1] for r in b)
print("step: %d kernels, %.3f ms wall, %.3f ms summed kernel time" % (len(b), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in b) / 1e6))
ev = sorted([(r[0], 1, r[2]) for r in b] + [(r[1], -1, r[2]) for r in b])
depth, last, cover = 0, t0, defaultdict(float)
alone = defaultdict(float)
active = []
for t, d, name in ev:
    cover[min(depth, 3)] += t - last
    if depth == 1 and active:
        alone[active[0]] += t - last
    last = t
    if d == 1:
        active.append(name)
    else:
        active.remove(name)
    depth += d
tot = t1 - t0
print("idle %.1f%%  one kernel %.1f%%  two %.1f%%  three+ %.1f%%" % tuple(100 * cover[i] / tot for i in range(4)))
print("time with exactly one kernel resident, by kernel:")
for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
    print("   %-50s %.3f ms" % (k, v / 1e6))
