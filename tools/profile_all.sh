#!/bin/bash
# The round's rocprofv3 evidence in one call (run on the GPU box through gpurun):
#   bash tools/profile_all.sh r05            -> gpurun_out/r05_<name>_{stats,fetch,write}/ + r05_<name>_stats_bench.json
# then, back in the authoring container:  bash tools/profile_all.sh --summarize r05
# (profiles/summarize.py per configuration: kernel statistics CSV + PMC summary into profiles/, traffic into
#  profiles/pmc_traffic.json keyed by workload and stamped with the hash of the kernel sources it was collected on).
# Kernel statistics in one pass, FETCH_SIZE / WRITE_SIZE each in its own --pmc pass (tools/profile.sh).
CONFIGS=(
  "upper|cfg3_float32_upper|"
  "sym|cfg3_float32_sym|--symmetric"
  "int8u|cfg3_int8_upper|--ld-dtype int8"
  "mix_upper|cfg3_float32_upper_mixture4|--model mixture"
  "mix|cfg3_float32_sym_mixture4|--symmetric --model mixture"
  "grid_upper|cfg3_float32_upper_grid32|--model grid"
  "grid|cfg3_float32_sym_grid32|--symmetric --model grid"
  "f64|cfg3_int8_upper_f64|--ld-dtype int8 --precision float64"
  "fast|cfg3_float32_upper_fast|--math fast"
  "fast_int8u|cfg3_int8_upper_fast|--math fast --ld-dtype int8"
)
HERE="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "--summarize" ]; then
  TAG=$2
  for c in "${CONFIGS[@]}"; do IFS='|' read -r name key args <<< "$c"
    t=${TAG}_${name}
    [ -d $HERE/gpurun_out/${t}_stats ] || { echo "skip $t (not collected)"; continue; }
    python3 $HERE/profiles/summarize.py $t $HERE/gpurun_out/${t}_stats $HERE/gpurun_out/${t}_fetch $HERE/gpurun_out/${t}_write $key > /dev/null && echo "summarised $t -> $key"
    cp $HERE/gpurun_out/${t}_stats_bench.json $HERE/profiles/${t}_bench_under_rocprof.json
  done
  exit 0
fi
TAG=$1
for c in "${CONFIGS[@]}"; do IFS='|' read -r name key args <<< "$c"
  echo "== ${TAG}_${name}: $args"
  bash $HERE/tools/profile.sh ${TAG}_${name} $args | tail -4 | cut -c1-200
done
