#!/bin/bash
# One BASELINE configuration on the GPU: the -m gpu tests matching a pattern, then its bench line.
# Usage on the GPU box: bash tools/check_config.sh <config> <pytest -k pattern> [bench args]
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT" || exit 1
cfg=$1; pat=$2; shift 2
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$pat" 2>&1 | tail -3
for rep in 1 2; do
timeout -k 10 300 python3 bench.py --config $cfg --no-cpu-baseline --no-secondary --steps 1000 "$@" 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("config %s: %.0f samples/s %.1f us/step whole-step %.3f | compose %.1f us per launch (alone %.1f) raster %.1f geom %.1f" % (d["config"]["baseline_config"], d["value"], d["ms_per_step"]*1e3, d["roofline"]["whole_step_frac"], d["roofline"]["kernel_ms"]*1e3, d["roofline"]["kernel_ms_alone"]*1e3, d["kernel_ms_alone"]["raster"]*1e3, d["kernel_ms_alone"]["geom"]*1e3))'
done
