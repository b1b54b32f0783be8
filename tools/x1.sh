cd $GRAFT_REPO_ROOT
for cfg in "3 4" "4 8" "5 8" "6 8" "3 8"; do
  set -- $cfg
  OFDG_CHAINS=$1 GPU_MAX_HW_QUEUES=$2 python3 bench.py --steps 1500 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chains $1 hwq $2: %.0f samples/s  %.1f us/step  compose %.1f us frac %.3f whole %.0f GB/s' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['hbm_gbs_whole_step']))"
done
