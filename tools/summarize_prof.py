"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into profiles/<name>.md."""
import csv, sys, os, collections

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]

def stats(path):
    rows = list(csv.DictReader(open(path)))
    return [(short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])) for r in rows]

def counters(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg

if __name__ == "__main__":
    d, out = sys.argv[1], sys.argv[2]
    lines = ["| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for n, c, t, a, p in stats(os.path.join(d, "prof_stats", "bench_kernel_stats.csv"))[:12]:
        lines.append(f"| {n} | {c} | {t:.2f} | {a:.3f} | {p:.2f} |")
    lines += ["", "| kernel | counter | launches | sum | per launch (KiB) |", "|---|---|---|---|---|"]
    for sub in ("prof_fetch", "prof_write"):
        p = os.path.join(d, sub, "bench_counter_collection.csv")
        if not os.path.exists(p):
            continue
        agg = counters(p)
        for (n, cn), (c, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
            lines.append(f"| {n} | {cn} | {c} | {v:.4g} | {v / c:.4g} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
