cd $GRAFT_REPO_ROOT
for ch in 2 3 4 5 6 8; do
  for rep in 1 2 3; do
    OFDG_CHAINS=$ch python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chains $ch steps 20: %.0f samples/s  %.1f us/step' % (d['value'], d['ms_per_step']*1e3))"
  done
done
