#!/bin/bash
# compose-alone timing of several libs on one box: tools/alone.sh label=lib[,ENV=V] ...
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for arm in "$@"; do
  label=${arm%%=*}; rest=${arm#*=}
  lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  [ "$lib" = "-" ] && lib=$GRAFT_REPO_ROOT/optical-flow-2d-data-generation_amd/lib/libofdg.so
  echo "[$r] $label: $(env OFDG_LIB=$lib $envs WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1 | sed 's/.*geom=/geom=/')"
done
done
