# PMC passes over one gemm shape, hand-written kernel and the library's:  bash tools/pmc_gemm.sh N K
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
N=${1:-256}; K=${2:-544}; T=r05_pmc_gemm_${N}_${K}
GEMM_ONE_LIBRARY=1 bash tools/pmc_cmd.sh $T python $GRAFT_REPO_ROOT/tools/gemm_one.py $N $K > /dev/null
PMC_GLOB="gpurun_out/$T/pmc_*/p_counter_collection.csv" python tools/pmc_any.py k_gemm Cijk > gpurun_out/r05/pmc_gemm_${N}_${K}.txt
rm -rf gpurun_out/$T
cat gpurun_out/r05/pmc_gemm_${N}_${K}.txt
