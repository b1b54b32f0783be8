#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/s9_tests.log 2>&1; tail -3 gpurun_out/r03/s9_tests.log
b=$(timeout -k 10 200 python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench %.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f; raster %.1f / %.1f geom %.1f / %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['kernel_ms_alone']*1e3, d['kernel_ms']['raster']*1e3, d['kernel_ms_alone']['raster']*1e3, d['kernel_ms']['geom']*1e3, d['kernel_ms_alone']['geom']*1e3))")
c=$(timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('20 steps: %.0f' % (d['value']))")
echo "$b | $c"
PASS_TIMEOUT=120 bash tools/pmc_compose.sh r03/pmc_cold > gpurun_out/r03/s9_pmc_cold.txt 2>&1
PASS_TIMEOUT=120 bash tools/pmc_compose.sh r03/pmc_warm POOLN=48 > gpurun_out/r03/s9_pmc_warm.txt 2>&1
PASS_TIMEOUT=120 bash tools/pmc_compose.sh r03/pmc_bgonly BGONLY=1 > gpurun_out/r03/s9_pmc_bgonly.txt 2>&1
tail -60 gpurun_out/r03/s9_pmc_cold.txt
