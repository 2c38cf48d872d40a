#!/bin/bash
# Build an instrumented / experimental variant of the library that differs from the main build only in the PANEL translation
# units (estep_panel.h, launch_panel.inc; with --grid: estep_grid_mfma.h, launch_grid.inc): compile those with the extra flags,
# link with the main objects.
#   tools/build_variant.sh trace -DVIPRS_SWEEP_TRACE      -> build/libviprs_hip_trace.so   (load through VIPRS_HIP_LIB)
set -e
cd "$(dirname "$0")/../viprs_amd/csrc"
name=$1; shift
fam=panel
if [ "$1" = "--grid" ]; then fam=grid; shift; fi          # (--grid: the batched grid kernel's translation units instead)
obj=../../build/obj_$name; mkdir -p $obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-pass-failed"
for f in ${fam}_f32 ${fam}_i8 ${fam}_i16; do /opt/rocm/bin/hipcc $FLAGS -DVIPRS_EXPERIMENTAL "$@" -c $f.hip -o $obj/$f.o & done; wait
others=$(ls ../../build/obj/*.o | grep -v "/${fam}_")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $others $obj/${fam}_f32.o $obj/${fam}_i8.o $obj/${fam}_i16.o -ldl -o ../../build/libviprs_hip_$name.so
echo built build/libviprs_hip_$name.so
