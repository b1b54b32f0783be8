#!/bin/bash
# Round profile: bench line + rocprofv3 kernel stats + HBM traffic counters (separate --pmc passes).
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
: ${GRAFT_REPO_ROOT:?}  # (set by gpurun; refuse to run from an unknown place)
tag=${1:-r01}
out=$PWD/gpurun_out/$tag
mkdir -p $out
python3 bench.py > $out/bench_line.json 2> $out/bench_stderr.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $out/bench_line_profiled.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 60 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 60 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py $out/trace > $out/kernel_stats.txt
python3 tools/pmcstats.py $out/pmc_fetch > $out/pmc_fetch_size.txt
python3 tools/pmcstats.py $out/pmc_write > $out/pmc_write_size.txt
tail -n 12 $out/kernel_stats.txt; cat $out/pmc_fetch_size.txt $out/pmc_write_size.txt | grep -A3 compose; cat $out/bench_line.json
