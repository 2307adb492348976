#!/bin/bash
# Builds tools/exp/librt_<name>.so from the tree's sources with extra -D flags (for tools/exp_libs.sh A/B runs):
#   tools/build_variant.sh <name> [-DFLAG=value ...]          the PRODUCT's objects (one kernel, no test entry points)
#   DIAG=1 tools/build_variant.sh <name> [-DFLAG ...]         the diagnostic library's objects (-DRT_DIAG_VARIANTS)
name=$1; shift
cd "$(dirname "$0")/../raytracing_c_amd/csrc" || exit 1
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wno-unused-function ${RAFLAGS--mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -amdgpu-prealloc-sgpr-spill-vgprs}"      # RAFLAGS= for the build without the Makefile's register-allocation flags
tmp=$(mktemp -d)
srcs="rt_kernels"; [ -n "$DIAG" ] && { srcs="rt_kernels rt_kernels_diag rt_wavefront"; F="$F -DRT_DIAG_VARIANTS"; }
objs=""
for f in $srcs; do /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o $tmp/$f.o || exit 1; objs="$objs $tmp/$f.o"; done
/opt/rocm/bin/hipcc $F "$@" -c rt_api.cpp -o $tmp/rt_api.o || exit 1
make -s rt_denoise.o rt_build.o rt_scene_build.o
mkdir -p ../../tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/librt_$name.so $objs $tmp/rt_api.o rt_denoise.o rt_build.o rt_scene_build.o -lpthread
rm -rf $tmp; ls -la ../../tools/exp/librt_$name.so
