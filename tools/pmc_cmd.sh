#!/bin/bash
# rocprofv3 passes over an arbitrary command (run on the GPU box via gpurun): tools/pmc_cmd.sh <tag> <command...>
# CSVs land in gpurun_out/<tag>/; same recipe as pmc_passes.sh (counters in their own runs, --kernel-trace only, each under timeout).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- "$@" > $OUT/stats.log 2>&1
rm -f $OUT/stats/s_kernel_trace.csv
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- "$@" > $OUT/pmc_$i.log 2>&1
  echo "pass $i ($set): rc=$?"
  rm -f $OUT/pmc_$i/p_kernel_trace.csv
done
