# rocprofv3 kernel statistics of the band kernel on one banded component (run on the GPU box):
#   gpurun -- 'bash tools/prof_banded.sh [banded_bench args]'  ->  gpurun_out/banded_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tmp_banded
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmp_banded -- python3 $R/tools/banded_bench.py "$@" > $R/gpurun_out/banded_bench_under_rocprof.txt 2>&1
cat $R/gpurun_out/tmp_banded/*/*kernel_stats.csv | head -8 | cut -c1-200 | tee $R/gpurun_out/banded_kernel_stats.csv
tail -2 $R/gpurun_out/banded_bench_under_rocprof.txt
