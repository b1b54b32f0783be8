#!/bin/bash
set -e
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for w in 0 1 2; do
  a=$(OFDG_WARM=$w WARM=16 ITERS=96 timeout -k 10 120 python3 tools/exp_compose.py 2>&1 | tail -1 | sed 's/.*geom=/geom=/')
  b=$(OFDG_WARM=$w timeout -k 10 200 python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench %.0f samples/s %.1f us/step (compose %.1f us in pipeline)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
  c=$(OFDG_WARM=$w timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('20 steps: %.0f' % (d['value']))")
  echo "[$rep] OFDG_WARM=$w: alone $a | $b | $c"
done
done
