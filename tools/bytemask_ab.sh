#!/bin/bash
# experiment: configs[2]'s step with BYTE-PLANE masks for the point-item levels of the table gradient
# (MaskPlan::fine_kind 3, cmp_block_bytes) against the nibble planes (UCN_BWD_BYTE_MASKS=0); =2: every point-item level
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for s in ${UCN_AB_SET:-0 1 2}; do
  export UCN_BWD_BYTE_MASKS=$s
  bash tools/train_top.sh > /dev/null 2>&1; mv gpurun_out/train_top.txt gpurun_out/r06/train_top_bytemask$s.txt
  python tools/train_prof.py 2>&1 | tail -1 | cut -c1-60 > gpurun_out/r06/train_ms_bytemask$s.txt
done
for s in ${UCN_AB_SET:-0 1 2}; do echo "== UCN_BWD_BYTE_MASKS=$s: $(cat gpurun_out/r06/train_ms_bytemask$s.txt)"; grep "total kernel\|k_march_features\|k_cast_cache" gpurun_out/r06/train_top_bytemask$s.txt | cut -c1-150; done
