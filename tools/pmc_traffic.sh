#!/bin/bash
# HBM traffic of the compose kernel for one BASELINE configuration: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes over `bench.py --config N` (counter collection serialises the kernels).  Usage on the GPU box, from the repo
# root: bash tools/pmc_traffic.sh <config> ; prints "<config> <kernel> <fetch_kb_raw> <write_kb_raw> <dur_us>"
: ${GRAFT_REPO_ROOT:?}  # (set by gpurun; refuse to run from an unknown place)
cfg=${1:-2}
out=$PWD/gpurun_out/pmc_c$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --steps 60 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --steps 60 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmcstats.py $out/fetch compose > $out/fetch.txt
python3 tools/pmcstats.py $out/write compose > $out/write.txt
cat $out/fetch.txt $out/write.txt
