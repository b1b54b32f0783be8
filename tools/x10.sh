cd $GRAFT_REPO_ROOT
for rg in 512 128 256 1024; do
  for run in "20 5" "2000 50"; do
    set -- $rg $run
    OFDG_RASTER_GRID=$1 python3 bench.py --steps $2 --warmup $3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('raster grid $1 x4, steps $2: %.0f samples/s  %.1f us/step  raster %.1f us co-running, %.1f alone' % (d['value'], d['ms_per_step']*1e3, d['kernel_ms']['raster']*1e3, d['kernel_ms_alone']['raster']*1e3))"
  done
done
