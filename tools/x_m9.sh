#!/bin/bash
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for rep in 1 2; do for v in ofdg ofdg_noobj ofdg_nobg ofdg_noboth; do
echo -n "[$rep] $v: "; OFDG_LIB=$L/lib$v.so MODE=9 WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1
done; done
