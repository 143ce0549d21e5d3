#!/bin/bash
# Unit counters (TA / TCP / TD / TCC, separate passes, --kernel-trace only) of the TRAINING step's gather-side kernels: the forward gather on
# incoherent rays, the mask pass and the table gradient.   bash tools/pmc_units_train.sh  ->  gpurun_out/r06/pmc_units_train.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r06_units_train; mkdir -p $OUT $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/train_prof.py $1 > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
PMC_GLOB="gpurun_out/r06_units_train/pmc_*/p_counter_collection.csv" python tools/pmc_any.py "k_march_features" "k_cast_cache" > gpurun_out/r06/pmc_units_train.txt
rm -rf $OUT
cat gpurun_out/r06/pmc_units_train.txt
