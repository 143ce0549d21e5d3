"""Per-kernel means of every counter in rocprofv3 PMC CSVs (any kernel name):  PMC_GLOB=... python tools/pmc_any.py [substring ...]"""
import csv, glob, collections, os, sys
agg = collections.defaultdict(lambda: [0, 0.0]); dur = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(os.environ.get("PMC_GLOB", "gpurun_out/pmc_*/p_counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        if sys.argv[1:] and not any(s in n for s in sys.argv[1:]):
            continue
        agg[(n, r["Counter_Name"])][0] += 1; agg[(n, r["Counter_Name"])][1] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[n][0] += 1; dur[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(dur):
    print(f"== {k}: avg {dur[k][1] / dur[k][0] / 1e6:.3f} ms over {dur[k][0]} profiled launches")
    for (n, c), (cnt, v) in sorted(agg.items()):
        if n == k:
            print(f"   {c:32s} {v / cnt:14.4g}")
