#!/bin/bash
# Ordered kernel list of the LAST training step of tools/train_prof.py (rocprofv3 kernel trace): which launches sit
# between the HIP nodes, to map the glue to source regions.  Writes gpurun_out/train_seq.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/trainseq; mkdir -p gpurun_out
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trainseq -o ts -- python tools/train_prof.py $1 > /tmp/trainseq.log 2>&1
python - <<PY
import csv, glob, re
f = glob.glob("/tmp/trainseq/**/ts_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
adam = [i for i, e in enumerate(ev) if "k_adam_step(" in e[2]]
# a step ends with its last k_adam_step (2 per step: both tables) + the foreach ops of the small parameters
last = adam[-1]; prev = adam[-3]
step = ev[prev + 1:last + 1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void at::native::", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:150]
out = open("gpurun_out/train_seq${1:+_$1}.txt", "w")
t0 = step[0][0]
for s, e, n in step:
    out.write(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  {short(n)}\n")
out.write(f"kernels {len(step)} span {(step[-1][1] - t0) / 1e6:.3f} ms busy {sum(e - s for s, e, _ in step) / 1e6:.3f} ms\n")
PY
tail -1 gpurun_out/train_seq${1:+_$1}.txt
