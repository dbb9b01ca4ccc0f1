#!/usr/bin/env python3
"""Per-kernel LDS bank-conflict rate from a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE` pass: conflict cycles per
LDS-active cycle (MI355X_MICROARCH.md, LDS).  usage: lds_conflict_summary.py <pmc dir> <workload label> > profiles/rNN_lds_conflicts.csv"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("mspi::", "").split("(")[0][:70]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
print("workload,kernel,dispatches,avg_SQ_LDS_BANK_CONFLICT,avg_SQ_LDS_IDX_ACTIVE,conflict_cycles_per_lds_active_cycle")
rows = []
for k, cs in acc.items():
    if "SQ_LDS_BANK_CONFLICT" not in cs or "SQ_LDS_IDX_ACTIVE" not in cs:
        continue
    c, a = cs["SQ_LDS_BANK_CONFLICT"], cs["SQ_LDS_IDX_ACTIVE"]
    conf, act = c[0] / c[1], a[0] / a[1]
    if act <= 0:
        continue
    rows.append((act * a[1], k, a[1], conf, act, conf / act))
for _, k, n, conf, act, frac in sorted(rows, reverse=True)[:16]:
    print("%s,%s,%d,%.0f,%.0f,%.4f" % (sys.argv[2], k.replace(",", ";"), n, conf, act, frac))
