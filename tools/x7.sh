cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for rep in 1 2; do
for arm in "noprio $L/libofdg_noprio.so" "prio $L/libofdg.so"; do
  set -- $arm
  for run in "20 5" "20 5" "2000 50"; do
    set -- $arm $run
    OFDG_LIB=$2 python3 bench.py --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 steps $3: %.0f samples/s  %.1f us/step  compose %.1f us (alone %.1f) prep alone %.1f+%.1f co-running %.1f+%.1f' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['kernel_ms_alone']*1e3, d['kernel_ms_alone']['geom']*1e3, d['kernel_ms_alone']['raster']*1e3, d['kernel_ms']['geom']*1e3, d['kernel_ms']['raster']*1e3))"
  done
done
done
