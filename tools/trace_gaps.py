#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: busy spans, idle gaps > 2 ms and what ran around them."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
t0 = ev[0][0]
end = ev[0][1]
gaps = []
for i, (s, e, name, q, st) in enumerate(ev[1:], 1):
    if s - end > 2_000_000:
        gaps.append((end - t0, s - end, i))
    end = max(end, e)
print("%d kernels over %.1f ms; %d idle gaps > 2 ms" % (len(ev), (end - t0) / 1e6, len(gaps)))
for at, dur, i in gaps[-12:]:
    print("  gap of %.1f ms at t=%.1f ms; before: %s (queue %s) | after: %s (queue %s)" % (dur / 1e6, at / 1e6, ev[i - 1][2], ev[i - 1][3], ev[i][2], ev[i][3]))
# longest kernels
top = sorted(ev, key=lambda x: x[1] - x[0], reverse=True)[:8]
for s, e, name, q, st in top:
    print("  long kernel %.2f ms at t=%.1f ms: %s (queue %s)" % ((e - s) / 1e6, (s - t0) / 1e6, name, q))
