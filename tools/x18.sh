#!/bin/bash
# cold-start staggering of the chains' first compose kernels: the driver's short run (20 steps after 5 warm-up steps) and a long run
cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5 6; do
for arm in "OFDG_STAGGER=0" "OFDG_STAGGER=1"; do
  b=$(env $arm python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "[$r] $arm 20 steps: $b"
done; done
for r in 1 2; do
for arm in "OFDG_STAGGER=0" "OFDG_STAGGER=1"; do
  b=$(env $arm python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))")
  echo "[$r] $arm 1500 steps: $b"
done; done
