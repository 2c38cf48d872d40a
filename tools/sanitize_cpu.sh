#!/bin/bash
# CPU sanitizer runs (the GPU pool offers neither device ASan nor XNACK):
#   1. the oracle's C restatement under ASan + UBSan (gcc) on the golden / reference-comparison tests,
#   2. the HOST side of libviprs_hip.so (planner, plan / state bookkeeping, C ABI argument handling) under ASan
#      (-Xarch_host: device code is compiled as usual) on the CPU tests that go through the C ABI.
# Nothing is installed or left in the tree: builds go to $OUT (default /tmp/viprs_san).
#   bash tools/sanitize_cpu.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/viprs_san}
mkdir -p $OUT
cd $R
gcc -O1 -g -std=c11 -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    oracle/estep_oracle.c -o $OUT/liboracle.so -lm
cp oracle/liboracle.so $OUT/liboracle.orig.so
trap 'cp $OUT/liboracle.orig.so $R/oracle/liboracle.so' EXIT
cp $OUT/liboracle.so oracle/liboracle.so
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests/test_golden.py tests/test_oracle_vs_ref.py tests/test_synthetic.py -x -q -m "not gpu"
cp $OUT/liboracle.orig.so oracle/liboracle.so
make -C viprs_amd/csrc OBJDIR=$OUT/obj OUT=$OUT/libviprs_hip_asan.so \
    EXTRA_CXXFLAGS="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -g"
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
VIPRS_HIP_LIB=$OUT/libviprs_hip_asan.so ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT \
    python -m pytest tests/test_planner.py tests/test_abi.py tests/test_zarr_ld.py tests/test_parallel.py tests/test_synth_device.py -x -q -m "not gpu"
