cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/r01_stats $R/gpurun_out/r01_fetch $R/gpurun_out/r01_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01_stats -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $R/gpurun_out/r01_stats_bench.json 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01_fetch -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r01_write -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-seconds 0 > /dev/null 2>&1
ls $R/gpurun_out/r01_fetch/*/ | head
head -3 $R/gpurun_out/r01_fetch/*/*counter_collection.csv
cat $R/gpurun_out/r01_stats/*/*kernel_stats.csv | head -8
cat $R/gpurun_out/r01_stats_bench.json
