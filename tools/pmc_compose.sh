#!/bin/bash
# PMC passes over the compose kernel running ALONE (tools/exp_compose.py: resident batches, kernels serialised on one
# stream).  Usage on the GPU box: bash tools/pmc_compose.sh <tag> [env assignments for the python script...]
# Each pass is its own rocprofv3 --pmc run (no tracing domains combined with counters).
: ${GRAFT_REPO_ROOT:?}  # (set by gpurun; refuse to run from an unknown place)
tag=${1:-pmc}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export "$@" 2>/dev/null
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 5 ${PASS_TIMEOUT:-150} rocprofv3 --pmc $line --output-format csv -d $out/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/exp_compose.py > $out/p$i.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $out/p$i compose >> $out/summary.txt 2>> $out/err.txt
done < <(if [ -n "$PASSFILE" ]; then cat $GRAFT_REPO_ROOT/$PASSFILE; else cat <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_IFETCH SQ_CYCLES
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_GATE_EN1_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_REQ_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
PASSES
fi)
cat $out/summary.txt
grep -h "compose=" $out/p*.log | head -3
