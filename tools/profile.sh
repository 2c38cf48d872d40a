#!/bin/bash
# rocprofv3 collection for profiles/ (run on the GPU box through gpurun):
#   bash tools/profile.sh <tag> [bench.py args]     e.g.  bash tools/profile.sh r02 --low-memory
# then, back in the authoring container:
#   python profiles/summarize.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write <traffic-key>
# Counters are collected in their own passes (kernel-trace only), one PMC counter per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/${TAG}_stats $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-secondary "$@" > $R/gpurun_out/${TAG}_stats_bench.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary "$@" > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-secondary "$@" > /dev/null 2>&1
cat $R/gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -8 | cut -c1-170
cut -c1-300 $R/gpurun_out/${TAG}_stats_bench.json
