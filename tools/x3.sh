cd $GRAFT_REPO_ROOT
for g in 0 4096 8192 3072; do
  echo "grid $g: $(BGONLY=1 OFDG_COMPOSE_GRID=$g WARM=16 ITERS=96 python3 tools/exp_compose.py 2>&1 | tail -1)"
done
