#!/bin/bash
: ${GRAFT_REPO_ROOT:?}
L=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib
REPS=2 STEPS=1000 bash $GRAFT_REPO_ROOT/tools/ab.sh base=- noflow=$L/libofdg_abl11.so nobilerp=$L/libofdg_abl12.so neither=$L/libofdg_abl13.so
echo "background-only batches:"
cd $GRAFT_REPO_ROOT
for lib in libofdg.so libofdg_abl13.so; do echo "$lib: $(OFDG_LIB=$L/$lib BGONLY=1 WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1)"; done
echo "pool of 48:"
for lib in libofdg.so libofdg_abl13.so; do echo "$lib: $(OFDG_LIB=$L/$lib POOLN=48 WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1)"; done
