#!/bin/bash
# tools/run_ab.sh <out name> "<bench args>" <label>=<lib name or -> ...   (one gpurun call: optional tests of the in-tree library, then tools/ab_bench.sh)
: ${GRAFT_REPO_ROOT:?}; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
out=$1; args=$2; shift 2
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
if [ -n "$TESTS" ]; then timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$TESTS" > gpurun_out/r06/${out}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06/${out}_tests.log; fi
arms=()
for a in "$@"; do l=${a%%=*}; f=${a#*=}; [ "$f" = "-" ] || f=$L/libofdg_$f.so; arms+=("$l=$f"); done
bash tools/ab_bench.sh "$args" "${arms[@]}" 2>&1 | tee gpurun_out/r06/$out.txt
