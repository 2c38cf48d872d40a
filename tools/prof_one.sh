cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tmp_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmp_stats -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 "$@" > /dev/null 2>&1
cat $R/gpurun_out/tmp_stats/*/*kernel_stats.csv | head -6 | cut -c1-230
