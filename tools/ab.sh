#!/bin/bash
# A/B on ONE box: tools/ab.sh "<label>=<lib path or ->[,ENV=V...]" ...   ("-" = the in-tree libofdg.so)
# each arm: compose alone (tools/exp_compose.py) + bench.py, repeated REPS times interleaved
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
REPS=${REPS:-2}
for r in $(seq $REPS); do
for arm in "$@"; do
  label=${arm%%=*}; rest=${arm#*=}
  lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  [ "$lib" = "-" ] && lib=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg.so
  a=$(env OFDG_LIB=$lib $envs WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1 | sed 's/.*geom=/geom=/')
  b=$(env OFDG_LIB=$lib $envs python3 bench.py --steps ${STEPS:-1500} --no-cpu-baseline --no-secondary ${BENCH_ARGS:---background-prep 0} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench %.0f samples/s %.1f us/step (compose %.1f us in pipeline)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
  echo "[$r] $label: alone $a | $b"
done
done
