"""Per-kernel totals from a rocprofv3 --kernel-trace CSV (kernel_trace.csv), one line per (kernel, grid size).
usage: python tools/kernel_times.py <dir or csv> [name filter]"""
import collections
import csv
import os
import re
import sys


def find_csv(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit("no kernel_trace.csv under " + path)


def main(argv):
    rows = list(csv.DictReader(open(find_csv(argv[0]))))
    flt = argv[1] if len(argv) > 1 else ""
    agg = collections.OrderedDict()
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if flt not in name:
            continue
        key = (name.split("(")[0][-60:], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(key, [0, 0.0, 1e30, []])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3].append(dur)
    for (name, grid), (n, tot, mn, durs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        durs.sort()
        print("%-62s grid %10d n=%3d avg %10.1f us min %10.1f us  median %10.1f max %10.1f" %
              (name, grid, n, tot / n, mn, durs[len(durs) // 2], durs[-1]))


if __name__ == "__main__":
    main(sys.argv[1:])
