#!/bin/bash
# GPU idle gaps inside bench.py's training step: rocprofv3 kernel trace of tools/train_prof.py, last step only
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/traingap
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/traingap -o tg -- python tools/train_prof.py > /tmp/traingap.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/traingap/**/tg_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the last optimiser step = after the last-but-one k_adam_step... take the final 360 kernels
ev = ev[-352:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
print(f"last ~step: span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms over {len(ev)} kernels")
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if s1 > e0:
        gaps.append(((s1 - e0) / 1e3, n0[:60], n1[:60]))
gaps.sort(reverse=True)
for g, a, b in gaps[:14]:
    print(f"{g:8.1f} us  after {a:60s} before {b}")
print("gaps > 20 us:", sum(1 for g in gaps if g[0] > 20), " total of those ms:", sum(g[0] for g in gaps if g[0] > 20) / 1e3)
PY
