"""Idle time between the kernels of the timed frame: reads a rocprofv3 --kernel-trace CSV of `python bench.py --no-train --no-extras
--no-cpu-baseline --steps 2 --warmup 1` and prints, per frame-sized window, wall time, summed kernel time and the gaps by size.
   rocprofv3 --kernel-trace -d gpurun_out/kt -o kt -- python bench.py --no-train --no-extras --no-cpu-baseline --steps 2 --warmup 1
   python tools/frame_gaps.py gpurun_out/kt"""
import csv, glob, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt"
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
assert files, root
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed frames: runs of kernels that start with the first k_march_features<..., true> after a long pause; simply take the last
# 2 x (number of NeRF-level launches per frame) region: find the NeRF-level gather launches and split them into frames of 240
idx = [i for i, r in enumerate(rows) if "k_march_features<" in r[2] and ", false>(" in r[2]]
per_frame = 240
frames = len(idx) // per_frame
print(f"{len(rows)} kernels, {len(idx)} NeRF-level gather launches = {frames} frames")
for fr in range(frames):
    lo = idx[fr * per_frame]
    hi = idx[(fr + 1) * per_frame - 1]
    # extend to the proposal-level kernels in front and the MLP / composite behind: previous / next gap above 2 ms bounds the frame
    a = lo
    while a > 0 and rows[a][0] - rows[a - 1][1] < 2_000_000 and (fr == 0 or a - 1 > idx[fr * per_frame - 1]):
        a -= 1
    b = hi
    while b + 1 < len(rows) and rows[b + 1][0] - rows[b][1] < 2_000_000 and (fr == frames - 1 or b + 1 < idx[(fr + 1) * per_frame]):
        b += 1
    seg = rows[a:b + 1]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"frame {fr}: {len(seg)} kernels, wall {wall:.2f} ms, kernel time {busy:.2f} ms, idle {sum(pos) / 1e6:.2f} ms in {len(pos)} gaps "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3:.1f} us, > 20 us: {sum(1 for g in pos if g > 20000)}, > 100 us: {sum(1 for g in pos if g > 100000)}, "
          f"largest {max(pos) / 1e3:.0f} us)")
    big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:6]
    for g, i in big:
        print(f"    {g / 1e3:8.1f} us after {seg[i][2][:60]}  before {seg[i + 1][2][:60]}")
