#!/bin/bash
# experiment: configs[2]'s step with the batch's rays visited in pixel-tile order instead of random order
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for s in 0 1; do
  export UCN_BENCH_SORT_RAYS=$s
  bash tools/train_top.sh > /dev/null 2>&1; mv gpurun_out/train_top.txt gpurun_out/r06/train_top_sort$s.txt
  python tools/train_prof.py 2>&1 | tail -1 | cut -c1-60 > gpurun_out/r06/train_ms_sort$s.txt
done
for s in 0 1; do echo "== UCN_BENCH_SORT_RAYS=$s: $(cat gpurun_out/r06/train_ms_sort$s.txt)"; grep "total kernel\|k_march_features\|k_cast_cache" gpurun_out/r06/train_top_sort$s.txt | cut -c1-150; done
