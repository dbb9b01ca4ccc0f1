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


def profiler_name(k):
    """rocprofv3 kernel name -> the name bench.py's HIP-event profiler uses (engine.conv)."""
    import re
    m = re.match(r"conv_gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        bm, bn, wm, wn, loader, prec = map(int, m.groups())
        return "conv_gemm<%d,%d,%s%s,%s>" % (bm, bn, "s" if loader else "v4", "w8" if wm * wn == 8 else "", "f16x3" if prec else "f32")
    m = re.match(r"conv_gemm_dma_kernel<(\d+), (\w+), (\w+), (\d+)(?:, (\w+))?>", k)
    if m:      # <BN, GATE, DENSE, NW[, APRE]>: rows = 32 * NW; APRE = pre-split activation planes
        return "conv_gemm<%d,%s,%s,f16x3>" % (32 * int(m.group(4)), m.group(1), "dma-presplit" if m.group(5) == "true" else "dma")
    m = re.match(r"conv_gemm_dma_kernel<(\d+),", k)
    if m:
        return "conv_gemm<128,%s,dma,f16x3>" % m.group(1)
    m = re.match(r"rowgemm_kernel<(\d+),", k)
    if m:
        return "rowgemm<%s,f16x3>" % m.group(1)
    if k.startswith("mlp_fused_kernel"):
        return "mlp_fused"
    if k.startswith("attn_f16x3_kernel") or k.startswith("attn_kernel"):
        return "attention"
    if k.startswith("dw_strip_kernel") or k.startswith("dw_tile_kernel") or k.startswith("dw_kernel<false"):
        return "dwconv"
    if k.startswith("layernorm_kernel"):
        return "layernorm"
    return None


def main(src, tag):
    here = os.path.dirname(os.path.abspath(__file__))
    trace = "trace_serial" if os.path.isdir(os.path.join(src, "trace_serial")) else "trace"   # MSPI_STREAMS=0: one launch at a time
    stats = (glob.glob(os.path.join(src, trace, "*_kernel_stats.csv")) + glob.glob(os.path.join(src, trace, "*", "*_kernel_stats.csv")))[0]
    rows = list(csv.DictReader(open(stats)))
    pmc = {}
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(lambda: [0.0, 0])
        for f in glob.glob(os.path.join(src, sub, "*_counter_collection.csv")) + glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
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
    # per-launch HBM bytes keyed by bench.py's kernel names (weighted over the template variants that share a name)
    import json
    agg = {}
    for k in set(pmc["FETCH_SIZE"]) | set(pmc["WRITE_SIZE"]):
        pn = profiler_name(k)
        if pn is None:
            continue
        fe, wr = pmc["FETCH_SIZE"].get(k, [0.0, 0]), pmc["WRITE_SIZE"].get(k, [0.0, 0])
        a = agg.setdefault(pn, [0.0, 0, 0.0, 0])
        a[0] += 2 * fe[0] * 1024; a[1] += fe[1]; a[2] += wr[0] * 1024; a[3] += wr[1]
    wl = json.load(open(os.path.join(src, "workload.json"))) if os.path.exists(os.path.join(src, "workload.json")) else ["x3dl", 8, 224, 300]
    with open(os.path.join(here, "%s_traffic.json" % tag), "w") as f:
        json.dump({"workload": wl, "kernels": {k: {"read_bytes": a[0] / max(a[1], 1), "write_bytes": a[2] / max(a[3], 1), "launches": a[1]}
                                                for k, a in sorted(agg.items())}}, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
