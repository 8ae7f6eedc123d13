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

# ---- machine-readable HBM-side traffic of the ResBlock kernels (bench.py reads profiles/rNN_pmc_traffic.json) ------------
import json
import re


def logical(name):
    for kn, lg in (("k_lm_gemm", "ivf_lm_gemm"), ("k_lm_select", "ivf_lm_select"), ("k_lm_plan", "ivf_lm_plan"), ("k_coarse_gemm_ks", "ivf_coarse_gemm"),
                   ("k_coarse_pick", "ivf_coarse_pick"), ("k_scan_v", "ivf_scan_query_major"), ("k_post", "conv_post")):
        if kn in name:
            return lg
    m = re.search(r"k_upsI\w+?Li(\d+)E", name)
    if m:
        return "ups_c%s" % m.group(1)
    m = re.search(r"k_rb_(pair|full|stream)I\w+?Li(\d+)E", name)
    if not m:
        return None
    fam = {"pair": "rb_pair", "full": "rb_full", "stream": "rb_stream"}[m.group(1)]
    if fam == "rb_stream":  # template arguments <OpT, C, MI, NJ, NCO, ND, ...>: ND = 1 is the pair-level variant
        nums = re.findall(r"Li(\d+)E", name)
        if len(nums) >= 5 and nums[4] == "1":
            fam = "rb_stream1"
    return "%s_c%s" % (fam, m.group(2))


vals = defaultdict(dict)
for pm in sorted(glob.glob(os.path.join(root, "pmc*"))):
    files = glob.glob(os.path.join(pm, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        lg = logical(r["Kernel_Name"])
        if lg and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                                        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
            acc[(lg, r["Counter_Name"])] += float(r["Counter_Value"])
            n[(lg, r["Counter_Name"])] += 1
    for (lg, c), v in acc.items():
        vals[lg][c] = v / n[(lg, c)]
# average launch duration per logical kernel from the kernel-trace pass (for the effective clock)
dur = {}
if stats:
    acc_d, n_d = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(stats[0])):
        lg = logical(r["Name"])
        if lg:
            acc_d[lg] += float(r["TotalDurationNs"])
            n_d[lg] += int(r["Calls"])
    dur = {k: acc_d[k] / max(1, n_d[k]) for k in acc_d}
out = {"source": "rocprofv3 --pmc passes of `bench.py --steps 5 --warmup 2 --graph 0` (tools/profile.sh); per-dispatch averages; "
                 "bytes_per_launch = FETCH_SIZE KB x 2 (gfx950 correction, MI355X_MICROARCH.md section HBM) + WRITE_SIZE KB",
       "kernels": {}}
for lg, d in sorted(vals.items()):
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        e = {"fetch_size_kb": d["FETCH_SIZE"], "fetch_correction": 2.0, "write_size_kb": d["WRITE_SIZE"],
             "bytes_per_launch": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0}
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"):
            if c in d:
                e[c] = d[c]
        if lg in dur:
            e["avg_launch_ns"] = dur[lg]
            if "GRBM_GUI_ACTIVE" in d and dur[lg] > 0:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / wall time = the clock the chip actually ran at under this kernel
                e["effective_clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / dur[lg]
            if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["GRBM_GUI_ACTIVE"] > 0:
                e["mfma_duty"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)  # busy cycles / (cycles x 1024 SIMDs)
        out["kernels"][lg] = e
if out["kernels"]:
    json.dump(out, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
    print("== wrote pmc_traffic.json for", sorted(out["kernels"]))
    print("== per kernel: bytes per launch (FETCH x2 + WRITE), effective clock, MFMA duty (the power story: a busy matrix pipe lowers the clock)")
    print("%-24s %12s %10s %10s %10s" % ("kernel", "MB/launch", "avg_us", "clock_GHz", "mfma_duty"))
    for lg, e in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("avg_launch_ns", 0)):
        print("%-24s %12.1f %10.1f %10s %10s" % (lg, e["bytes_per_launch"] / 1e6, e.get("avg_launch_ns", 0) / 1e3,
                                                 "%.2f" % e["effective_clock_ghz"] if "effective_clock_ghz" in e else "-",
                                                 "%.2f" % e["mfma_duty"] if "mfma_duty" in e else "-"))
