#!/bin/bash
# Engine clock and power while the headline pipeline runs: tools/exp_clocks.sh "<bench args>"  (rocm-smi sampled every 0.5 s beside bench.py)
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
args=${1:---background-prep 1}
python3 bench.py --no-cpu-baseline --no-secondary --steps 60000 $args > /tmp/clk_line.json 2>/dev/null &
pid=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.5
done
wait $pid
python3 -c "import json; d=json.load(open('/tmp/clk_line.json')); print('%.0f samples/s %.1f us/step' % (d['value'], d['ms_per_step']*1e3))"
echo "idle:"; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" | tr -s ' ' | tr '\n' ';'; echo
