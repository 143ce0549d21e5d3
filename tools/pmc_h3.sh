# PMC passes over one split-engine gemm shape:  bash tools/pmc_h3.sh N K  -> gpurun_out/r06/pmc_h3_N_K.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
N=${1:-256}; K=${2:-256}; T=r06_pmc_h3_${N}_${K}
bash tools/pmc_cmd.sh $T python $GRAFT_REPO_ROOT/tools/gemm_one.py $N $K > /dev/null
PMC_GLOB="gpurun_out/$T/pmc_*/p_counter_collection.csv" python tools/pmc_any.py k_gemm_h3 > gpurun_out/r06/pmc_h3_${N}_${K}.txt
rm -rf gpurun_out/$T
cat gpurun_out/r06/pmc_h3_${N}_${K}.txt
