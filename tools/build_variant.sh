#!/bin/bash
# experiment build of libsfnative.so with extra hipcc flags into build_var/<name>/ (git-ignored; travels with gpurun)
# usage: tools/build_variant.sh <name> [flags...]     then run with SF_LIB_PATH=build_var/<name>/libsfnative.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=build_var/$name; mkdir -p $out
srcs="conv_igemm conv_sp conv_wino convnext_mlp aux_kernels api lift_splat voxelize sparse_index eval_kernels pack $SF_EXTRA_SRCS"
pids=()
for s in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c streamingflow_amd/csrc/$s.hip -o $out/$s.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""; for s in $srcs; do objs="$objs $out/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libsfnative.so $objs
echo $out/libsfnative.so
