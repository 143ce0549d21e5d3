#!/bin/bash
# rocprofv3 kernel stats of bench.py's training step, summed per category (run on the GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/trainprof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trainprof -o tp -- python tools/train_prof.py > /tmp/trainprof.log 2>&1
tail -1 /tmp/trainprof.log | cut -c1-60
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/trainprof/**/tp_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
cat=collections.Counter(); cnt=collections.Counter()
for r in rows:
    n=r["Name"]; t=float(r["TotalDurationNs"])/1e6/8; c=int(r["Calls"])/8
    if "Cijk" in n: k="gemm"
    elif "k_march_features_bwd" in n: k="feat_bwd"
    elif "k_cast_cache" in n: k="feat_bwd_cast"
    elif "k_march_features" in n: k="feat_fwd"
    elif "anonymous namespace)::k_" in n and "at::" not in n: k="own_other"
    elif "reduce_kernel" in n: k="reduce"
    elif "Cat" in n: k="cat"
    elif "elementwise" in n or "Elementwise" in n: k="elementwise"
    elif "copyBuffer" in n or "fill" in n.lower(): k="copy/fill"
    else: k="other"
    cat[k]+=t; cnt[k]+=c
for k,v in cat.most_common(25): print(f"{k:20s} {v:7.3f} ms/step {cnt[k]:6.1f} launches/step")
print("sum", round(sum(cat.values()),3), sum(cnt.values()))
PY
