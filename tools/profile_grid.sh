#!/bin/bash
# rocprofv3 collection for the batched grid E-step (run on the GPU box through gpurun):
#   bash tools/profile_grid.sh <tag> <workload> <n_models> [grid_bench.py args]
# Kernel statistics in one pass, FETCH_SIZE / WRITE_SIZE each in their own --pmc pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
for k in stats fetch write; do rm -rf $R/gpurun_out/${TAG}_$k; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/tools/grid_bench.py "$@" --reps 10 > $R/gpurun_out/${TAG}_bench.txt 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/grid_bench.py "$@" --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/grid_bench.py "$@" --reps 3 > /dev/null 2>&1
cat $R/gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -6 | cut -c1-200
cat $R/gpurun_out/${TAG}_bench.txt
