#!/bin/bash
# rocprofv3 kernel trace of bench.py's training step: the individual launches of the featurisation kernels (forward /
# masks / backward, NeRF-level and proposal-level call separately), last step.   -> gpurun_out/train_calls.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/traincalls; mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/traincalls -o tc -- python tools/train_prof.py $1 > /tmp/traincalls.log 2>&1
python - > gpurun_out/train_calls.txt <<PY
import csv,glob
f=glob.glob("/tmp/traincalls/**/tc_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_march_features" in r["Kernel_Name"] or "k_cast_cache" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows[-12:]:
    print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8}  {r['Kernel_Name'][:70]}")
PY
cat gpurun_out/train_calls.txt
