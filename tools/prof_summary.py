#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time (kernel trace) and per-kernel
averages of every collected PMC counter."""
import csv, glob, os, sys, collections

out = sys.argv[1]


def short(name):
    n = name
    for k in ("sweep_items_kernel", "sweep_long_rows_kernel", "gram_chunk_kernel", "gram_finalize_kernel",
              "sum_groups_kernel", "sum_to_double_kernel"):
        if k in n:
            return k
    return "other:" + n.replace("void ", "")[:48]


tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(tr[0])):
        agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print("== kernel trace (all dispatches, warm-up included) ==")
    print("%-28s %8s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "max_us", "pct"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-28s %8d %12.1f %12.1f %12.1f %6.1f%%" % (k, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, max(v) / 1e3,
                                                          100.0 * sum(v) / tot))
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not cc:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(cc[0])):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== %s: per-dispatch averages ==" % os.path.basename(d))
    for k in sorted(agg):
        if k.startswith("other:"):
            continue
        print("  " + k + "  " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())) +
              "  (n=%d)" % len(next(iter(agg[k].values()))))
