cd $GRAFT_REPO_ROOT
for g in 0 2048 4096 6144 8192 12288; do
  echo "== compose grid $g"
  OFDG_COMPOSE_GRID=$g WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1
  OFDG_COMPOSE_GRID=$g python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   bench: %.0f samples/s  %.1f us/step  compose %.1f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
done
