#!/bin/bash
# One rocprofv3 --pmc pass over bench.py for one library build: tools/pmc_kernel.sh <tag> "<counters>" "<bench args>" [lib] [kernel filter]
# -> prints the per-kernel averages (tools/pmcstats.py); counter collection serialises the kernels.
: ${GRAFT_REPO_ROOT:?}
tag=$1; counters=$2; args=$3; lib=${4:-$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg.so}; flt=${5:-ofdg}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && OFDG_LIB=$lib timeout -k 5 200 rocprofv3 --pmc $counters --output-format csv -d $out/p -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 $args > $out/log.txt 2>&1 )
python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $out/p $flt
rm -rf $out/p
