cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
run() { n=$1; shift; rm -rf gpurun_out/pm$n; ONLY256=1 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/pm$n -- python tools/conv_bench.py bf16 "final.resblock 256" > gpurun_out/pm$n.log 2>&1; }
run 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS
run 2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM
run 3 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run 4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
python tools/pmc_report.py glds_kernel gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3 gpurun_out/pm4
