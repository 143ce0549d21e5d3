"""Aggregate gpurun_out/pmc_*/p_counter_collection.csv per (kernel, counter): mean per launch."""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
dur = collections.defaultdict(lambda: [0, 0.0])
import os
for f in sorted(glob.glob(os.environ.get("PMC_GLOB", "gpurun_out/pmc_*/p_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not n.startswith("k_"):
            continue
        agg[(n, r["Counter_Name"])][0] += 1
        agg[(n, r["Counter_Name"])][1] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[n][0] += 1
            dur[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
want = sys.argv[1:] or sorted({k for k in dur if k.startswith(("k_march_features<", "k_field_mlp_h", "k_field_mlp<"))})
want = [k for w in want for k in sorted(dur) if k == w or k.startswith(w + "<")]
for k in want:
    if dur[k][0] == 0:
        continue
    print(f"== {k}: avg duration {dur[k][1] / dur[k][0] / 1e6:.3f} ms over {dur[k][0]} profiled launches")
    for (n, c), (cnt, v) in sorted(agg.items()):
        if n == k:
            print(f"   {c:40s} {v / cnt:16.4g} per launch")
