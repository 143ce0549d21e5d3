#!/bin/bash
# rocprofv3 PMC passes over bench.py's training step (tools/train_prof.py [heads | R] as $2); same recipe as pmc_passes.sh.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-pmc_train}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/train_prof.py $2 > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($set): rc=$?"
  rm -f $OUT/pmc_$i/p_kernel_trace.csv
done
cd $R && PMC_GLOB="gpurun_out/$TAG/pmc_*/p_counter_collection.csv" python tools/pmc_table.py k_march_features_bwd_cmp k_cast_cache_masks k_march_features k_train_fwd k_train_bwd > $OUT/table.txt 2>&1
find $OUT -name "p_counter_collection.csv" -delete
cat $OUT/table.txt | head -120
