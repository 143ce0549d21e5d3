#!/bin/bash
# Round-5 evidence run on the GPU box: full -m gpu suite, the default bench line, kernel stats + PMC passes, training-step profiles.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r05c
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r05c/gpu_suite.txt
python bench.py > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err
bash tools/pmc_passes.sh r05c_pmc > gpurun_out/r05c/pmc_passes.log 2>&1
PMC_GLOB="gpurun_out/r05c_pmc/pmc_*/p_counter_collection.csv" python tools/pmc_table.py > gpurun_out/r05c/pmc_table.txt 2>&1
python tools/traffic_json.py r05c_pmc > gpurun_out/r05c/traffic.log 2>&1; cp gpurun_out/r05c_pmc/traffic.json gpurun_out/r05c/traffic.json
cp $(find gpurun_out/r05c_pmc/stats -name "s_kernel_stats.csv" | head -1) gpurun_out/r05c/kernel_stats.csv
find gpurun_out/r05c_pmc -name "p_counter_collection.csv" -delete; rm -rf gpurun_out/r05c_pmc/stats
bash tools/train_top.sh > /dev/null 2>&1; mv gpurun_out/train_top.txt gpurun_out/r05c/train_top.txt
bash tools/train_top.sh heads > /dev/null 2>&1; mv gpurun_out/train_top_heads.txt gpurun_out/r05c/train_top_heads.txt
bash tools/train_top.sh fp32 > /dev/null 2>&1; mv gpurun_out/train_top_fp32.txt gpurun_out/r05c/train_top_fp32.txt
bash tools/train_top.sh heads_fp32 > /dev/null 2>&1; mv gpurun_out/train_top_heads_fp32.txt gpurun_out/r05c/train_top_heads_fp32.txt
bash tools/pmc_train.sh r05c_pmctrain > gpurun_out/r05c/pmc_train.log 2>&1; cp gpurun_out/r05c_pmctrain/table.txt gpurun_out/r05c/pmc_table_train.txt
python tools/gemm_f32_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r05c/gemm_f32_bench.txt
tail -3 gpurun_out/r05c/gpu_suite.txt; head -c 600 gpurun_out/r05c/bench.json
python tools/train_ops.py 2>&1 | grep -v amdgpu > gpurun_out/r05c/train_ops.txt
