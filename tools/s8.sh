#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/s8_tests.log 2>&1; tail -3 gpurun_out/r03/s8_tests.log
echo "alone: $(WARM=16 ITERS=96 timeout -k 10 120 python3 tools/exp_compose.py 2>&1 | tail -1)"
for rep in 1 2; do
b=$(timeout -k 10 200 python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench %.0f samples/s %.1f us/step (compose %.1f us in pipeline, alone %.1f; raster %.1f / %.1f geom %.1f / %.1f)' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['kernel_ms_alone']*1e3, d['kernel_ms']['raster']*1e3, d['kernel_ms_alone']['raster']*1e3, d['kernel_ms']['geom']*1e3, d['kernel_ms_alone']['geom']*1e3))")
c=$(timeout -k 10 100 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('20 steps: %.0f' % (d['value']))")
echo "$b | $c"
done
cd /tmp && export TMPDIR=/tmp
WARM=16 ITERS=96 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/s8_alone -o t -- python3 $GRAFT_REPO_ROOT/tools/exp_compose.py > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r03/s8_alone/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][:40]
    if any(k in n for k in ('compose','layer','raster','geom','sample')): print('  %-40s calls %5s avg %8.1f us min %8.1f max %8.1f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
