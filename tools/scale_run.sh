#!/bin/bash
# The multi-GPU scaling run of bench.py on ONE node (what the driver does at round end):
#   tools/scale_run.sh [max_gpus] [steps] [warmup]      ->  gpurun_out/scale/bench_N.json for N = 1, 2, 4, ... max_gpus
# One process per GPU over RCCL (torch.distributed backend "nccl"); rank 0 prints the JSON line, whose `value` is the
# whole-job rays/s (the same frame on N ranks: strong scaling, the packed all-gather inside the timed region).
# Efficiency = value_N / (N * value_1) -- computed by whoever reads the files, not here.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MAX=${1:-8}; STEPS=${2:-3}; WARM=${3:-1}
OUT=$R/gpurun_out/scale; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
N=1
while [ "$N" -le "$MAX" ] && [ "$N" -le "$HAVE" ]; do
  if [ "$N" -eq 1 ]; then
    timeout 900 python "$R/bench.py" --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-extras > "$OUT/bench_$N.json" 2> "$OUT/bench_$N.err"
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      "$R/bench.py" --gpus "$N" --steps "$STEPS" --warmup "$WARM" > "$OUT/bench_$N.json" 2> "$OUT/bench_$N.err"
  fi
  echo "N=$N rc=$? $(python -c "import json,sys; r=json.load(open('$OUT/bench_$N.json')); print(r['value'], r['ms_per_step'])" 2>/dev/null)"
  N=$((N * 2))
done
