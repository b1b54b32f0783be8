#!/bin/bash
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for rep in 1 2; do for v in ${VARIANTS}; do
echo -n "[$rep] $v: "; OFDG_LIB=$L/libofdg_$v.so python3 bench.py --config 3 --no-cpu-baseline --no-reference-equivalent --steps 1000 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("config 3: %.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f)" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["kernel_ms"]*1e3, d["roofline"]["kernel_ms_alone"]*1e3))'
done; done
