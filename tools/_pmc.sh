cd /tmp && export TMPDIR=/tmp
for d in 0 17; do
export OFDG_DBG=$d
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d /root/repo/gpurun_out/pmca$d -o p -- python3 /root/repo/tools/exp_compose.py > /dev/null 2>&1
rocprofv3 --pmc SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY SPI_CSN_WAVE SPI_RA_RES_STALL_CSN SPI_RA_SGPR_SIMD_FULL_CSN --output-format csv -d /root/repo/gpurun_out/pmcb$d -o p -- python3 /root/repo/tools/exp_compose.py > /dev/null 2>&1
echo "dbg $d"; python3 /root/repo/tools/pmcstats.py /root/repo/gpurun_out/pmca$d 2>&1 | grep -A8 compose_kernel;  python3 /root/repo/tools/pmcstats.py /root/repo/gpurun_out/pmcb$d 2>&1 | grep -A9 compose_kernel
done
