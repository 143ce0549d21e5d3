#!/bin/bash
# rocprofv3 kernel stats of bench.py's training step: the top kernels by time per step (8 steps incl. warm-up).
#   tools/train_top.sh [heads]  -> gpurun_out/train_top[_heads].txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/trainprof; mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trainprof -o tp -- python tools/train_prof.py $1 > /tmp/trainprof.log 2>&1
tail -1 /tmp/trainprof.log | cut -c1-200
python - > gpurun_out/train_top${1:+_$1}.txt <<PY
import csv,glob
f=glob.glob("/tmp/trainprof/**/tp_kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6/8
print(f"total kernel time {tot:.3f} ms/step, {sum(int(r['Calls']) for r in rows)/8:.0f} launches/step")
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6/8:8.3f} ms/step {int(r['Calls'])/8:7.1f} calls  {r['Name'][:150]}")
PY
cat gpurun_out/train_top${1:+_$1}.txt
