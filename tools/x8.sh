cd $GRAFT_REPO_ROOT
for cfg in "0 4" "1 4" "1 3" "1 5"; do
  set -- $cfg
  for run in "20 5" "20 5" "2000 50"; do
    set -- $cfg $run
    OFDG_HOSTSCHED=$1 OFDG_CHAINS=$2 timeout -k 5 120 python3 bench.py --steps $3 --warmup $4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('hostsched $1 chains $2 steps $3: %.0f samples/s  %.1f us/step  compose %.1f us (alone %.1f) in flight %.2f' % (d['value'], d['ms_per_step']*1e3, r['kernel_ms']*1e3, r['kernel_ms_alone']*1e3, r['launches_in_flight']))"
  done
done
