"""Per-kernel averages of the counters in a rocprofv3 --pmc CSV (counter_collection.csv).
usage: python tools/pmc_kernels.py <dir> [name filter]"""
import collections
import csv
import os
import re
import sys


def main(argv):
    path = None
    for root, _, files in os.walk(argv[0]):
        for f in files:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(root, f)
    if path is None:
        raise SystemExit("no counter_collection.csv under " + argv[0])
    flt = argv[1] if len(argv) > 1 else ""
    agg = collections.OrderedDict()
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-48:]
        if flt not in name:
            continue
        key = (name, int(r["Grid_Size"]))
        agg.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], 0.0)
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key].add(r["Dispatch_Id"])
    for key, ctr in agg.items():
        n = len(disp[key])
        print("%-50s grid %9d n=%d  " % (key[0], key[1], n) + "  ".join("%s=%.4g" % (c, v / n) for c, v in ctr.items()))


if __name__ == "__main__":
    main(sys.argv[1:])
