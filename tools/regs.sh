#!/bin/bash
# tools/regs.sh <kernel name regex> [extra hipcc flags]: registers / scratch / occupancy of matching kernels (compile only)
pat=$1; shift
cd "$(dirname "$0")/../optical-flow-2d-data-generation_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 "$@" -c csrc/ofdg_api.hip -o /tmp/regs_x.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "Name: _ZN4ofdg[0-9]*\($pat\)" | grep -E "Name|VGPRs:|Scratch|Occupancy" | sed "s/.*remark: //; s/ \[-Rpass.*//; s/Function Name: _ZN4ofdg[0-9]*\([a-z0-9_]*\)E.*/\1/" | paste - - - -
