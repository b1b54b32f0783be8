#!/bin/bash
# texture prefetch in geom_kernel (OFDG_PREFETCH=1, default) against none (=0): configs 2 and 5, long and short runs
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for arm in "OFDG_PREFETCH=0" "OFDG_PREFETCH=1"; do
for c in 2 5; do
  b=$(env $arm python3 bench.py --config $c --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f; geom alone %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('kernel_ms_alone',0)*1e3, d.get('kernel_ms_alone',{}).get('geom',0)*1e3))")
  echo "[$r] $arm config $c: $b"
done; done; done
