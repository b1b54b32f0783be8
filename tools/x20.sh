#!/bin/bash
# what would a free sampler be worth?  OFDG_X_SKIP_SAMPLER=1: the chains' records of their first batches are rendered again and again (timing only)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for arm in "OFDG_X_SKIP_SAMPLER=0" "OFDG_X_SKIP_SAMPLER=1"; do
  b=$(env $arm python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "[$r] $arm: $b"
done; done
