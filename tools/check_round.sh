#!/bin/bash
# GPU check of the tree: the -m gpu tests, then the bench line the driver takes (--steps 20 --warmup 5) and the default one.
# Usage on the GPU box: bash tools/check_round.sh <tag>   -> gpurun_out/<tag>/{tests.log, bench_driver_like.json, bench.json}
: ${GRAFT_REPO_ROOT:?}
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-check}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; tail -4 $out/tests.log
show='import json,sys
d=json.loads(sys.stdin.read()); r=d.get("centre_crop_backgrounds") or d.get("reference_equivalent") or {}
print("background_prep %d: %.0f samples/s %.1f us/step whole-step %.3f | compose %.1f us per launch (alone %.1f) | background_prep %s: %s | %s" % (d["config"]["background_prep"], d["value"], d["ms_per_step"]*1e3, d["roofline"]["whole_step_frac"], d["roofline"]["kernel_ms"]*1e3, d["roofline"]["kernel_ms_alone"]*1e3, r.get("background_prep", "-"), ("%.0f" % r["value"]) if r else "-", d["config"]["context"]))'
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_like.json 2> $out/bench_driver_like.err && python3 -c "$show" < $out/bench_driver_like.json
timeout -k 10 400 python3 bench.py ${BENCH_ARGS:-} > $out/bench.json 2> $out/bench.err && python3 -c "$show" < $out/bench.json
