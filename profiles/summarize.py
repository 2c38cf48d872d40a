#!/usr/bin/env python3
"""Condenses rocprofv3 output (written under gpurun_out/ on the GPU box) into the small summaries kept
in profiles/.  Usage: python profiles/summarize.py <round-tag> <stats-dir> <fetch-dir> <write-dir> <traffic-key>

HBM bytes follow MI355X_MICROARCH.md (HBM / rocprofv3 PMC section): FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts exactly half
of the bytes of wide coalesced streaming reads, so   hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import sys

tag, stats_dir, fetch_dir, write_dir, key = sys.argv[1:6]
here = os.path.dirname(os.path.abspath(__file__))


def counters(d, name):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)     # (the newest run of the directory)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and ("viprs::estep" in r["Kernel_Name"] or "viprs::tile_f64" in r["Kernel_Name"]):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


stats = max(glob.glob(os.path.join(stats_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(stats))]
with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    w.writerows(rows)

fetch, write = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
calls = {r["Name"].split("(")[0].replace("void ", ""): int(r["Calls"]) for r in rows}
per_kernel = {}
sweeps = None
for k in sorted(set(fetch) | set(write)):
    n = len(fetch.get(k, [])) or len(write.get(k, []))
    per_kernel[k] = {
        "launches_profiled": n,
        "FETCH_SIZE_KiB_per_launch": sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [1]))),
        "WRITE_SIZE_KiB_per_launch": sum(write.get(k, [0])) / max(1, len(write.get(k, [1]))),
    }
    per_kernel[k]["hbm_bytes_per_launch"] = int(
        (2 * per_kernel[k]["FETCH_SIZE_KiB_per_launch"] + per_kernel[k]["WRITE_SIZE_KiB_per_launch"]) * 1024)
# launches per sweep: the panel kernel of the smallest class runs exactly once per sweep
ref = min((len(v) for v in fetch.values()), default=1)
total = 0
for k, v in per_kernel.items():
    per_sweep = len(fetch.get(k, [])) / ref
    v["launches_per_sweep"] = per_sweep
    total += v["hbm_bytes_per_launch"] * per_sweep
summary = {"tag": tag, "key": key, "hbm_bytes_per_sweep": int(total), "kernels": per_kernel,
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes (MI355X_MICROARCH.md)"}
json.dump(summary, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1)
# bench.py quotes the figure only while the sources of the kernel family it was collected on are unchanged
sys.path.insert(0, os.path.dirname(here))
from viprs_amd.utils import kernel_id  # noqa: E402
tp = os.path.join(here, "pmc_traffic.json")
traffic = json.load(open(tp)) if os.path.exists(tp) else {}
fam = kernel_id.family_of(key)
traffic[key] = {"hbm_bytes_per_sweep": int(total), "family": fam, "src_hash": kernel_id.source_hash(fam),
                "kernels": sorted(per_kernel), "collected": tag}
json.dump(traffic, open(tp, "w"), indent=1)
print(json.dumps(summary, indent=1))
