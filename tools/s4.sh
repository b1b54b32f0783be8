#!/bin/bash
set -e
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg_stamps.so
for w in 0 2; do
for m in "pipeline" "alone"; do
echo "OFDG_WARM=$w"
OFDG_WARM=$w OFDG_LIB=$V timeout -k 10 200 python3 tools/exp_stamps.py $m 2>&1 | grep -v amdgpu.ids | cut -c1-250
done
done
