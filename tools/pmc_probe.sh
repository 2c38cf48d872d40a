#!/bin/bash
# Development tool: a few TCP / SQ counters of the sweep kernel (separate passes, kernel-trace only; the TA_* / GRBM_*
# counters did not collect on this pool -- the pass runs into its timeout):
#   bash tools/pmc_probe.sh [bench.py args]     -> prints per-counter averages over the profiled launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  rm -rf $R/gpurun_out/pmcprobe
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcprobe -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary "$@" > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmcprobe/*/*counter_collection.csv")
agg = collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if "estep_sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(f"{k:45s} {sum(v) / len(v):.4g}")
PY
done
