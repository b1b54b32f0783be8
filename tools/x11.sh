#!/bin/bash
# composite masks: division-free u/255.f (one Newton step) vs the divisions: configs 5, 4 and 2, interleaved
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for r in 1 2; do
for arm in base newton; do
for c in 5 4 2; do
  b=$(env OFDG_LIB=$L/libofdg_$arm.so python3 bench.py --config $c --steps 1000 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('kernel_ms_alone',0)*1e3))")
  echo "[$r] $arm config $c: $b"
done; done; done
