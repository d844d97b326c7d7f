#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel time (kernel trace) and per-kernel
averages of every collected PMC counter."""
import csv, glob, os, sys, collections

out = sys.argv[1]


def short(name):
    n = name
    for k in ("sweep_stream_kernel", "short_rows_kernel", "whiten_factor_kernel", "gram_bound_kernel", "scores_split_kernel", "topk_rows_kernel", "select_candidates_kernel", "wide_row_items_kernel",
              "sweep_items_kernel", "sweep_long_rows_kernel", "gram_chunk_kernel", "gram_finalize_kernel",
              "sum_groups_kernel", "sum_to_double_kernel", "cg_items_kernel", "cg_long_rows_kernel", "scores_kernel",
              "gram_wide_chunk_kernel", "sum_slots_kernel"):
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

# HBM traffic of one half-sweep's per-row kernels, for bench.py's roofline.traffic: the sum over the kernels between
# the taps of their traffic per half-sweep.  FETCH_SIZE and
# WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads (global_load and
# LDS-DMA alike - the sweep's gathers) at 64 bytes, so the read side is doubled; WRITE_SIZE is exact.  Infinity-Cache
# hits are counted in, not excluded.
import json
ROW_KERNELS = ("sweep_stream_kernel", "short_rows_kernel", "sweep_items_kernel", "sweep_long_rows_kernel", "sum_slots_kernel", "wide_row_items_kernel",
               "cg_items_kernel", "cg_long_rows_kernel")
fetch, write = {}, {}
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(d) else []
    for f in cc:
        vals = collections.defaultdict(lambda: collections.defaultdict(list))
        n_sum = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k in ROW_KERNELS:
                vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            elif k == "sum_to_double_kernel":          # two launches end every half-sweep (the |delta| sum)
                n_sum[row["Counter_Name"]] += 1
        # a kernel can run more than once per half-sweep (the stream kernel: the long rows' chunk items, then the
        # whole rows) or in some half-sweeps only (the long rows' kernels), so: total over its dispatches / number of
        # half-sweeps
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            have = {k: v[name] for k, v in vals.items() if name in v}
            if not have:
                continue
            n_half = n_sum[name] // 2 if n_sum[name] >= 2 else min(len(x) for x in have.values())
            for k, x in have.items():
                (fetch if name == "FETCH_SIZE" else write)[k] = sum(x) / n_half
if fetch and write:
    ft, wt = sum(fetch.values()), sum(write.values())
    t = {"kernels": sorted(fetch), "FETCH_SIZE_KiB_avg": ft, "WRITE_SIZE_KiB_avg": wt,
         "FETCH_SIZE_KiB_by_kernel": fetch, "WRITE_SIZE_KiB_by_kernel": write,
         "hbm_bytes_per_launch": ft * 1024 * 2 + wt * 1024,
         "note": "sum over the per-row kernels of a half-sweep of their traffic per half-sweep (user and item sweeps averaged); "
                 "read side = FETCH_SIZE x 2 (gfx950 correction)"}
    json.dump(t, open(os.path.join(out, "pmc_traffic_entry.json"), "w"), indent=1)
    print("== HBM traffic per half-sweep (%s): %.3f GB (fetch x2 %.3f GB + write %.3f GB)" %
          (", ".join(sorted(fetch)), t["hbm_bytes_per_launch"] / 1e9, ft * 2048 / 1e9, wt * 1024 / 1e9))
