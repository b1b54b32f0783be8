#!/bin/bash
# Per-kernel durations (rocprofv3 --kernel-trace --stats) of a bench.py run: tools/kernel_stats.sh <tag> "<bench args>" [lib]
# -> gpurun_out/<tag>_kernel_stats.txt
: ${GRAFT_REPO_ROOT:?}
tag=$1; args=$2; lib=${3:-$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg.so}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export OFDG_LIB=$lib
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary $args > $out/bench_line.json 2>/dev/null )
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py $out/trace > gpurun_out/${tag}_kernel_stats.txt
rm -rf $out/trace
head -8 gpurun_out/${tag}_kernel_stats.txt
