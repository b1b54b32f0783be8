#!/bin/bash
# mode 9 (config 3): where does compose_deform's time go?  ablations (WRONG pictures, timing only):
#   abl21 background never deforms | abl22 objects never deform | abl23 neither (the rigid path of the generic kernel)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for r in 1 2; do
for arm in base abl21 abl22 abl23; do
  b=$(env OFDG_LIB=$L/libofdg_$arm.so python3 bench.py --config 3 --steps 600 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('kernel_ms_alone',0)*1e3))")
  echo "[$r] $arm config 3: $b"
done; done
