cd /tmp && export TMPDIR=/tmp
for d in 33 1 0; do
export OFDG_DBG=$d
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /root/repo/gpurun_out/pmc$d -o p -- python3 /root/repo/tools/exp_compose.py > /dev/null 2>&1
echo "dbg $d"; python3 /root/repo/tools/pmcstats.py /root/repo/gpurun_out/pmc$d 2>&1 | grep -A9 "compose"
done
