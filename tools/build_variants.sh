#!/bin/bash
# Builds A/B variants of libbtx.so into build_variants/ (measurement only): tools/build_variants.sh NAME "-DFLAG ..." ...
set -e
cd "$(dirname "$0")/../bayesian_torch_amd/csrc"
mkdir -p ../../build_variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( d=$(mktemp -d); for f in btx_api btx_contract_f32 btx_contract_bf16 btx_patch_f32 btx_patch_bf16 btx_x3 btx_wgrad btx_bn; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. $flags -c $f.hip -o $d/$f.o & done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o ../../build_variants/libbtx_$name.so; rm -rf $d; echo built $name ) &
done
wait
