#!/bin/bash
# mode 9: grids of the deform kernels, one box; each arm "ENV=V,ENV=V"; prints step rate and the alone durations of a profiled pass
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for arm in "$@"; do
  envs=$(echo "$arm" | tr ',' ' ')
  b=$(env $envs python3 bench.py --config 3 --steps 600 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "[$r] $arm: $b"
done; done
