#!/bin/bash
# background_prep = 1: its -m gpu tests, kernel statistics of `bench.py --background-prep` (rocprofv3) and the step rate.
# Usage on the GPU box: bash tools/check_bgprep.sh
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "background_prep or small_pool or mixed or texture_list" 2>&1 | tail -2
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/bgprep_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --background-prep 1 --no-cpu-baseline --steps 400 > /dev/null 2>&1 )
python3 tools/kstats.py gpurun_out/r03/bgprep_trace | grep -E "bgprep|compose|raster|geom|sample" 
for rep in 1 2; do
timeout -k 10 300 python3 bench.py --background-prep 1 --no-cpu-baseline --steps 1000 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print("background_prep = 1: %.0f samples/s %.1f us/step whole-step %.3f" % (d["value"], d["ms_per_step"]*1e3, d["roofline"]["whole_step_frac"]))'
done
