#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/s7_tests.log 2>&1; tail -3 gpurun_out/r03/s7_tests.log
for la in 0 1 2 3; do
b=$(OFDG_LOOKAHEAD=$la timeout -k 10 200 python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench %.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f; raster %.1f / %.1f geom %.1f / %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['kernel_ms_alone']*1e3, d['kernel_ms']['raster']*1e3, d['kernel_ms_alone']['raster']*1e3, d['kernel_ms']['geom']*1e3, d['kernel_ms_alone']['geom']*1e3))")
c=$(OFDG_LOOKAHEAD=$la timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('20 steps: %.0f' % (d['value']))")
echo "lookahead $la: $b | $c"
done
