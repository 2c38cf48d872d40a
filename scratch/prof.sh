cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-seconds 0 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/prof2/*/*kernel_stats.csv | head -12
