#!/bin/bash
# GPU box: per-launch durations of the skeleton kernels for the two-canopy batch (configs[3]).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ctrace
timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ctrace -- python $R/tools/probe_canopy.py "" > /tmp/ctrace.log 2>&1
python - <<'PY' > $R/gpurun_out/r04_canopy_trace.txt
import csv, glob
f = glob.glob("/tmp/ctrace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    if n in ("k_sk_select", "k_sk_claim", "k_sk_helper_tables"):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if n != "k_sk_claim" or d > 30:
            print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  {n:22s} {d:10.1f} us  grid {r['Grid_Size_X']}")
PY
grep -v amdgpu.ids /tmp/ctrace.log >> $R/gpurun_out/r04_canopy_trace.txt
