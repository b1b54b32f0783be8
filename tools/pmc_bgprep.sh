#!/bin/bash
# PMC passes over the background-preparation kernels (bench.py --background-prep; counter collection serialises the kernels).
# Usage on the GPU box: bash tools/pmc_bgprep.sh  -> gpurun_out/r03/pmc_bgprep.txt
: ${GRAFT_REPO_ROOT:?}
out=$GRAFT_REPO_ROOT/gpurun_out/r03/pmc_bgprep
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $line --output-format csv -d $out/p$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --background-prep 1 --no-cpu-baseline --steps 40 --warmup 10 > $out/p$i.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $out/p$i bgprep >> $out/summary.txt 2>> $out/err.txt
  rm -rf $out/p$i
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
PASSES
cat $out/summary.txt > $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_bgprep.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r03/pmc_bgprep.txt
