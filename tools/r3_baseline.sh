#!/bin/bash
# Round-3 starting point on one box: single-cloud latency + stages, the driver's command (--steps 20), and a kernel trace of a
# single-cloud loop (launch count, kernel time per cloud).  usage: gpurun --timeout 900 -- 'bash tools/r3_baseline.sh <tag>'
tag=${1:-r3base}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
python $R/tools/time_single.py > $O/${tag}_single.txt 2>&1
ST_BENCH_MIN_UPTIME_S=20 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/${tag}_steps20.json 2> $O/${tag}_steps20.err
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_single -- python $R/tools/time_single.py 1000000 0.02 0 5 > $O/${tag}_prof_single.log 2>&1
cd $R
python - <<EOF | tee $O/${tag}_prof_single_summary.txt
import csv, glob
fs = glob.glob("$O/${tag}_prof_single/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
calls_per_cloud = 2 * (3 + 5 + 1)   # warm-up 3x2, loop 5x2, profiled pass 2
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = sum(int(r["Calls"]) for r in rows)
print("kernel time per cloud %.3f ms, launches per cloud %.0f" % (tot / calls_per_cloud / 1e6, n / calls_per_cloud))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%8.1f us/cloud %6.1f calls/cloud  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / calls_per_cloud / 1e3, int(r["Calls"]) / calls_per_cloud, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
EOF
