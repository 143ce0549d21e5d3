"""The first kernels of a timed frame with the idle time in front of each (same trace as tools/frame_gaps.py)."""
import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt"
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "k_march_features<" in r[2] and ", false>(" in r[2]]
start = idx[240 * 2 - 1] + 1          # behind the last NeRF-level gather of the second frame
for i in range(start, min(start + 130, len(rows))):
    gap = (rows[i][0] - rows[i - 1][1]) / 1e3
    print(f"{gap:9.1f} us idle | {(rows[i][1] - rows[i][0]) / 1e3:9.1f} us  {rows[i][2][:110]}")
