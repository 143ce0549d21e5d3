#!/bin/bash
# gemm_f32 A/B on the GPU box: correctness tests of csrc/gemm_f32.hip, then the shape bench per kernel variant
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r05
python -m pytest tests/test_dense_f32.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/dense_tests.txt
tail -3 gpurun_out/r05/dense_tests.txt
python tools/gemm_f32_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r05/gemm_f32_bench.txt
cat gpurun_out/r05/gemm_f32_bench.txt
for v in ${GEMM_VARIANTS:-1 2}; do
  echo "variant $v"; UCN_GEMM_VARIANT=$v python tools/gemm_f32_bench.py 2>&1 | grep "^gemm.*N 256 K \(256\|544\)" | tee gpurun_out/r05/gemm_f32_bench_variant$v.txt
  UCN_GEMM_VARIANT=$v python -m pytest tests/test_dense_f32.py -x -q -m gpu -k "gemm_f32_against" 2>&1 | tail -1
done
