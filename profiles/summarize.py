#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rN/{trace,pmc_fetch,pmc_write}) into the small, tracked
summaries under profiles/: per-kernel time (kernel_stats) and per-kernel HBM traffic per launch.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes
of a wide coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for
16-B-per-lane streaming stores."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("mspi::", "")
    return name.split("(")[0][:70]


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    pmc = {}
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    a = acc[short(r["Kernel_Name"])]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
        pmc[ctr] = acc
    out = os.path.join(here, "%s_kernel_summary.csv" % tag)
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,pct,hbm_read_MB_per_launch(2xFETCH_SIZE),hbm_write_MB_per_launch\n")
        for r in rows:
            k = short(r["Name"])
            fe, wr = pmc["FETCH_SIZE"].get(k), pmc["WRITE_SIZE"].get(k)
            f.write("%s,%s,%.3f,%.2f,%s,%s,%s\n" % (
                k.replace(",", ";"), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"],
                "%.3f" % (2 * fe[0] / fe[1] * 1024 / 1e6) if fe and fe[1] else "",
                "%.3f" % (wr[0] / wr[1] * 1024 / 1e6) if wr and wr[1] else ""))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
