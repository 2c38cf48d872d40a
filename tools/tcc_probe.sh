#!/bin/bash
# Development tool: L2 (TCC) request counters of the sweep kernel, one --pmc pass per group (kernel-trace only):
#   bash tools/tcc_probe.sh [bench.py args]     -> per-counter averages over the profiled launches
# (what FETCH_SIZE is made of: read requests to the fabric by size, L2 hits / misses, write-backs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf $R/gpurun_out/tccprobe
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/tccprobe -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/tccprobe/*/*counter_collection.csv")
agg = collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if "estep_sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
else:
    print("(no counter file for: $set)")
for k, v in agg.items():
    print(f"{k:40s} {sum(v) / len(v):.6g}")
PY
done
