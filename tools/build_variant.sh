#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> optical-flow-2d-data-generation_amd/lib/libofdg_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../optical-flow-2d-data-generation_amd"
mkdir -p build/v_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 "$@" -c csrc/ofdg_api.hip -o build/v_$name/ofdg_api.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "Name: _ZN4ofdg25compose_rigid_pow2" | grep -E "VGPRs:|Scratch|Occupancy" | sed "s/.*remark: /$name: /" | tr '\n' ' '; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o lib/libofdg_$name.so build/v_$name/ofdg_api.o build/realize.o build/sampler_ref.o build/layer.o build/warpfields.o build/comm.o -ldl
