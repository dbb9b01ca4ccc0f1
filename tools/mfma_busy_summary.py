#!/usr/bin/env python3
"""Per-kernel MFMA-pipe busy fraction from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
busy fraction = BUSY * 8 / (GUI_ACTIVE * 1024 SIMDs): GUI_ACTIVE is summed over the 8 XCDs, BUSY counts cycles per SIMD-MFMA
(32 per v_mfma_f32_32x32x16_f16).  usage: mfma_busy_summary.py <pmc dir> <workload label> > profiles/rNN_mfma_busy.csv"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("mspi::", "").split("(")[0][:70]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print("workload,kernel,dispatches,avg_SQ_VALU_MFMA_BUSY_CYCLES,avg_GRBM_GUI_ACTIVE(sum over 8 XCDs),mfma_pipe_busy_frac = BUSY*8/(GUI_ACTIVE*1024 SIMDs)")
rows = []
for k, cs in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    b, g = cs["SQ_VALU_MFMA_BUSY_CYCLES"], cs["GRBM_GUI_ACTIVE"]
    busy, gui = b[0] / b[1], g[0] / g[1]
    if busy <= 0:
        continue
    rows.append((busy * b[1], k, b[1], busy, gui, busy * 8 / (gui * 1024)))
rows.sort(reverse=True)
keep = rows[:14] + [r for r in rows[14:] if "attn" in r[1]]      # the attention kernels are a north-star target: always listed
for _, k, n, busy, gui, frac in keep:
    print("%s,%s,%d,%.0f,%.0f,%.4f" % (sys.argv[2], k.replace(",", ";"), n, busy, gui, frac))
