#!/bin/bash
# one batch of 24 at a time on one stream: solo kernel durations per cloud, by family (what DESIGN.md quotes)
R=$GRAFT_REPO_ROOT; tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo_$tag -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_solo_$tag.log 2>&1
cd $R
f=$(ls -t gpurun_out/prof_solo_$tag/*/*kernel_stats.csv | head -1)
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/prof_solo_$tag.log") if l.startswith("{")][-1])
print("bench line under the profiler: %.3f ms per step; warm-up steps run %d" % (d["ms_per_step"], d["config"]["warmup_steps_run"]))
# passes: warm-up + timed + upload + solo (24 clouds)
print("clouds", d["config"]["warmup_steps_run"] + 96 + 96 + 24)
open("gpurun_out/prof_solo_${tag}_clouds.txt", "w").write(str(d["config"]["warmup_steps_run"] + 96 + 96 + 24))
PY
python tools/family_summary.py $f $(cat gpurun_out/prof_solo_${tag}_clouds.txt) | tee gpurun_out/prof_solo_${tag}_families.txt
