#!/bin/bash
# rocprofv3 passes over one bench step (run on the GPU box via gpurun); CSVs land in gpurun_out/<tag>/.
# Pass 0 is the kernel trace + stats; the PMC passes follow the guide's recipe: counters in their own
# runs, --kernel-trace only.  Every pass runs under its own `timeout` (a TA_*/TCP_* pass once hung a box).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-train $BENCH_EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py $ARGS > $OUT/stats.log 2>&1
rm -f $OUT/stats/s_kernel_trace.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $R/bench.py $ARGS > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($set): rc=$?"
  rm -f $OUT/pmc_$i/p_kernel_trace.csv
done
ls $OUT/*/ | head -30
