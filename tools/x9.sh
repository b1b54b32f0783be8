cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
for rep in 1 2; do
for arm in "old $L/libofdg_old.so" "new $L/libofdg_cur.so"; do
  set -- $arm
  for run in "20 5" "20 5" "2000 50"; do
    set -- $arm $run
    OFDG_LIB=$2 python3 bench.py --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 steps $3: %.0f samples/s  %.1f us/step' % (d['value'], d['ms_per_step']*1e3))"
  done
done
done
cd /tmp && export TMPDIR=/tmp
for arm in "old $L/libofdg_old.so" "new $L/libofdg_cur.so"; do
  set -- $arm
  OFDG_LIB=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/samp_$1 -o t -- python3 $GRAFT_REPO_ROOT/tools/exp_counter.py > /dev/null 2>&1
  echo "$1: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/gpurun_out/samp_$1 | grep cs_sample)"
done
