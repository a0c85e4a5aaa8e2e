#!/usr/bin/env python3
"""Turns one tools/prof.sh output directory into the numbers DESIGN.md / bench.py quote:
average duration of the dominant kernel (rocprofv3 --stats) next to what bench.py measured in the
same run, and HBM traffic per launch from the two PMC passes, corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE x 2 for 16-B/lane streams on gfx950, WRITE_SIZE as is, both in KB)."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def counter(sub, name, kernel_sub):
    path = find(sub, "*counter_collection.csv")
    vals = []
    if path:
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == name and kernel_sub in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    return vals


bench = json.load(open(os.path.join(d, "bench_stats.json")))
stats = {}
path = find("stats", "*kernel_stats.csv")
with open(path) as f:
    for row in csv.DictReader(f):
        stats[row["Name"]] = row
dom = max((n for n in stats if "epoch_kernel" in n), key=lambda n: float(stats[n]["TotalDurationNs"]))
short = dom[dom.index("epoch_kernel"):dom.index("(", dom.index("epoch_kernel"))]
fetch = counter("fetch", "FETCH_SIZE", "epoch_kernel")
write = counter("write", "WRITE_SIZE", "epoch_kernel")
k, nnz = bench["config"]["k"], bench["config"]["nnz_per_gpu"]
out = {
    "workload": bench["config"]["workload"].split(" ")[0], "scale": bench["config"]["scale"], "k": k, "nnz": nnz,
    "kernel": short,
    "rocprof_calls": int(stats[dom]["Calls"]),
    "rocprof_avg_launch_us": float(stats[dom]["AverageNs"]) / 1e3,
    "bench_avg_launch_us_same_run": bench["roofline"]["avg_launch_us"],
    "updates_per_s_same_run": bench["value"],
    "FETCH_SIZE_KB_per_launch": sum(fetch) / max(1, len(fetch)),
    "WRITE_SIZE_KB_per_launch": sum(write) / max(1, len(write)),
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE tallies 64 B per 128-B request for 16-B/lane streams -> doubled; "
                  "WRITE_SIZE exact for 16-B/lane stores; x1024 (KB units); separate --pmc passes",
    "algorithmic_bytes_per_launch": (16 * k + 12) * nnz,
    "max_item_degree": bench["config"]["max_item_degree"],
    "sum_round_steps": bench["config"]["sum_round_steps"],
}
out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024
out["hbm_gbs"] = out["hbm_bytes_per_launch"] / (out["rocprof_avg_launch_us"] * 1e-6) / 1e9
out["other_kernels_us"] = {n[:60]: float(r["AverageNs"]) / 1e3 for n, r in stats.items()
                           if any(t in n for t in ("sse_kernel", "pack_kernel", "degree_kernel", "key_kernel"))}
print(json.dumps(out, indent=1))
