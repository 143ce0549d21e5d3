#!/bin/bash
# Round-6 evidence run on the GPU box: full -m gpu suite, the default bench line, kernel stats + PMC passes, training-step profiles
# (autocast and the fp32 route on both dense engines), GEMM benches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_suite.txt
python bench.py > $O/bench.json 2> $O/bench.err
bash tools/pmc_passes.sh r06c_pmc > $O/pmc_passes.log 2>&1
PMC_GLOB="gpurun_out/r06c_pmc/pmc_*/p_counter_collection.csv" python tools/pmc_table.py > $O/pmc_table.txt 2>&1
python tools/traffic_json.py r06c_pmc > $O/traffic.log 2>&1; cp gpurun_out/r06c_pmc/traffic.json $O/traffic.json
cp $(find gpurun_out/r06c_pmc/stats -name "s_kernel_stats.csv" | head -1) $O/kernel_stats.csv
find gpurun_out/r06c_pmc -name "p_counter_collection.csv" -delete; rm -rf gpurun_out/r06c_pmc/stats
for v in "" heads fp32 heads_fp32; do
  bash tools/train_top.sh $v > /dev/null 2>&1; mv gpurun_out/train_top${v:+_$v}.txt $O/train_top${v:+_$v}.txt
done
for v in fp32 heads_fp32; do
  UCN_F32_EXACT=1 bash tools/train_top.sh $v > /dev/null 2>&1; mv gpurun_out/train_top_$v.txt $O/train_top_${v}_exact_engine.txt
done
python tools/gemm_f32_bench.py 2>&1 | grep -v amdgpu > $O/gemm_bench.txt
python tools/train_fp32_ab.py 2>&1 | grep -v amdgpu > $O/train_fp32_ab.txt
bash tools/pmc_h3.sh 256 256 > /dev/null 2>&1; cp gpurun_out/r06/pmc_h3_256_256.txt $O/ 2>/dev/null
tail -3 $O/gpu_suite.txt; head -c 400 $O/bench.json; echo; cat $O/train_fp32_ab.txt
