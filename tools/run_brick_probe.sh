#!/bin/bash
# GPU box: tools/probe_brick_voxeliser.py plain and under rocprofv3 (per-kernel durations of both voxelisers)
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python tools/probe_brick_voxeliser.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_brick_voxeliser.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bvprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bvprof -- python $R/tools/probe_brick_voxeliser.py > /tmp/bvprof.log 2>&1
python - <<'PY' >> $R/gpurun_out/r04_brick_voxeliser.txt
import csv, glob
f = glob.glob("/tmp/bvprof/*/*kernel_stats.csv")[0]
print("rocprofv3 --kernel-trace --stats of the same script (average duration per launch):")
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if n.startswith("k_bv_") or n.startswith("k_vx_") or n.startswith("k_sort") or "cumsum" in r["Name"].lower() or "scan" in n.lower():
        print(f"  {float(r['AverageNs']) / 1e3:9.1f} us x {r['Calls']:>5s}  {n[:90]}")
PY
