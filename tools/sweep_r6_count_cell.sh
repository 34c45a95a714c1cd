#!/bin/bash
# GPU box: outlier_removal's counting search -- cap of its grid cell as a multiple of the MEAN bound (tuning pseudo code 101), after
# the advisor's fix (the mean is taken over the valid, non-NaN bounds: the cap no longer shrinks with the valid fraction).
cd $GRAFT_REPO_ROOT
for m in 45 36 28 22; do
  ST_SKELETON_PARAMS="101=$m" ST_BENCH_MIN_UPTIME_S=12 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('count cell cap $m/100 x mean bound: %.1f M points/s, %.4f ms/step; outlier stage %.4f ms/cloud in the set, %.3f ms single; single cloud %.2f ms' % (d['value']/1e6, d['ms_per_step'], d['stage_ms']['outlier_removal'], d['single_cloud']['stage_ms']['outlier_removal'], d['single_cloud']['ms']))"
done
