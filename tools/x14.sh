#!/bin/bash
# mode 9 (config 3): staged transformed textures; arms: lib variant [env]
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for r in 1 2; do
for arm in "$@"; do
  lib=${arm%%,*}; envs=""
  if [[ "$arm" == *,* ]]; then envs=$(echo "${arm#*,}" | tr ',' ' '); fi
  b=$(env OFDG_LIB=$L/$lib $envs python3 bench.py --config 3 --steps 600 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline'].get('kernel_ms_alone',0)*1e3))")
  echo "[$r] $arm: $b"
done; done
