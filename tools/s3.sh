#!/bin/bash
set -e
: ${GRAFT_REPO_ROOT:?}
cd /tmp && export TMPDIR=/tmp
for w in 0 1 2; do
  export OFDG_WARM=$w WARM=16 ITERS=96
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/warm$w -o t -- python3 $GRAFT_REPO_ROOT/tools/exp_compose.py > /dev/null 2>&1
  echo "OFDG_WARM=$w"
  python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/r03/warm$w/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][:40]
    if any(k in n for k in ('compose','warm','raster','geom')): print('  %-40s calls %5s avg %8.1f us min %8.1f max %8.1f' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
