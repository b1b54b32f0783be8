#!/bin/bash
set -e
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
V=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg_stamps.so
for m in "pipeline" "alone" "alone hold"; do
OFDG_LIB=$V timeout -k 10 200 python3 tools/exp_stamps.py $m 2>&1 | grep -v amdgpu.ids
done
export POOLN=48
for m in "pipeline" "alone"; do
echo "pool of 48 images (Infinity-Cache resident):"
OFDG_LIB=$V timeout -k 10 200 python3 tools/exp_stamps.py $m 2>&1 | grep -v amdgpu.ids
done
