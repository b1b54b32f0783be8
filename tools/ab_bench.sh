#!/bin/bash
# Same-box A/B of bench.py between library builds: tools/ab_bench.sh "<bench args>" <label>=<lib or -> ...   ("-" = the in-tree libofdg.so)
# REPS repetitions (default 2), arms interleaved.  Prints samples/s, us/step and the compose launch in the pipeline per arm.
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
args=$1; shift
for r in $(seq ${REPS:-2}); do
for arm in "$@"; do
  label=${arm%%=*}; lib=${arm#*=}
  [ "$lib" = "-" ] && lib=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg.so
  env OFDG_LIB=$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); ka=d['kernel_ms_alone']; kp=d['kernel_ms']; print('[$r] %-12s %8.0f samples/s %6.1f us/step (compose %.1f us in the pipeline, %.1f alone; geom %.1f raster %.1f alone; background_prep %.1f in the pipeline, %.1f alone)' % ('$label', d['value'], d['ms_per_step']*1e3, d['roofline']['per_launch']['kernel_ms']*1e3, d['roofline']['per_launch']['kernel_ms_alone']*1e3, ka['geom']*1e3, ka['raster']*1e3, kp.get('background_prep', 0)*1e3, ka.get('background_prep', 0)*1e3))"
done
done
