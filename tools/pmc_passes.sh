#!/bin/bash
# rocprofv3 PMC passes over one bench step (run on the GPU box via gpurun); CSVs land in gpurun_out/pmc_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-train $BENCH_EXTRA"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$i -o p -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_$i.log 2>&1
  rm -f $R/gpurun_out/pmc_$i/p_kernel_trace.csv
done
ls $R/gpurun_out/pmc_*/ | head -20
