"""Condense rocprofv3 output directories (kernel stats + PMC csv) into a short per-kernel table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.split("(")[0]
    for pre in ("void rvcmi::", "rvcmi::"):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:70]


stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    print("== kernel stats (%s)" % os.path.relpath(stats[0], root))
    rows = list(csv.DictReader(open(stats[0])))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print("%-72s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:24]:
        print("%-72s %8s %12.1f %10.2f %6s" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                             float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))
for pm in sorted(glob.glob(os.path.join(root, "pmc*"))):
    files = glob.glob(os.path.join(pm, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    names = sorted({c for k in agg for c in agg[k]})
    print("== PMC per dispatch average (%s)" % os.path.basename(pm))
    print("%-60s " % "kernel" + " ".join("%18s" % n[-18:] for n in names))
    for k in sorted(agg, key=lambda k: -max(agg[k].values())):
        print("%-60s " % k[:60] + " ".join("%18.4g" % (agg[k][n] / max(1, cnt[(k, n)])) for n in names))
