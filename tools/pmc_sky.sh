# PMC passes over the heads training step (serial sky), sky training kernels' counters -> gpurun_out/r05/pmc_sky.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05; OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_pmcsky; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $GRAFT_REPO_ROOT/tools/heads_prof_serial.py > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
PMC_GLOB="gpurun_out/r05_pmcsky/pmc_*/p_counter_collection.csv" python tools/pmc_any.py k_sky_train k_train_fwd k_train_bwd > gpurun_out/r05/pmc_sky.txt
rm -rf $OUT; cat gpurun_out/r05/pmc_sky.txt
