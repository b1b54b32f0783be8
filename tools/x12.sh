#!/bin/bash
# the driver's short run (20 steps after 5 warm-up steps): does an active wait in the runtime's synchronize change it?
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6; do
for arm in "A=0" "ROC_ACTIVE_WAIT_TIMEOUT=5000" "HIP_FORCE_DEV_KERNARG=1"; do
  b=$(env $arm python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "[$r] $arm: $b"
done; done
