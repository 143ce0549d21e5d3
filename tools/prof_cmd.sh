#!/bin/bash
# rocprofv3 kernel stats of an arbitrary command: tools/prof_cmd.sh <out-name> <divisor> <cmd...>  -> gpurun_out/<out-name>.txt (top kernels, ms per <divisor>)
name=$1; div=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_$$; mkdir -p gpurun_out/$(dirname $name)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$$ -o p -- "$@" > /tmp/prof_$$.log 2>&1
python - $div > gpurun_out/$name.txt <<PY
import csv,glob,sys
div=float(sys.argv[1])
f=glob.glob("/tmp/prof_$$/**/p_kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/1e6/div
print(f"total kernel time {tot:.3f} ms per unit, {sum(int(r['Calls']) for r in rows)/div:.0f} launches per unit")
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/1e6/div:9.3f} ms {int(r['Calls'])/div:8.1f} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:140]}")
PY
cp $(find /tmp/prof_$$ -name "p_kernel_stats.csv" | head -1) gpurun_out/$name.kernel_stats.csv
rm -rf /tmp/prof_$$
