#!/bin/bash
# GPU box: SSSP levels per frontier launch 4 (default so far) against 8 / 6 on the NETWORK graph (the headline), alternating runs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for p in "6=4" "6=8" "6=6"; do
  ST_SKELETON_PARAMS="$p" ST_BENCH_MIN_UPTIME_S=12 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$p] %.1f M points/s, %.4f ms/step (passes %s); skeleton kernels %.3f ms/cloud in the set; single cloud %.2f ms (skeleton kernels %.2f)' % (d['value']/1e6, d['ms_per_step'], d['passes']['ms_per_step'], d['stage_ms']['skeleton_kernels'], d['single_cloud']['ms'], d['single_cloud']['stage_ms']['skeleton_kernels']))"
done; done
