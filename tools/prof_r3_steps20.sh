#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s20x -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_s20x.log 2>&1
cd $R; rm -f gpurun_out/prof_s20x/*/*kernel_trace.csv
python - <<PY
import csv,glob
f=sorted(glob.glob("gpurun_out/prof_s20x/*/*kernel_stats.csv"))[-1]
rows=list(csv.DictReader(open(f)))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:8]:
    print("calls %6d avg %8.1f us  %s"%(int(r["Calls"]),float(r["AverageNs"])/1e3,r["Name"][:70]))
PY
