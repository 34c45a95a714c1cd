#!/bin/bash
# round 3: SSSP by regions (12=1; 6 = activations per launch) against the frontier launches: tests, one cloud, the driver's 20 steps, 384 steps
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_region.txt; : > $O
ST_SKELETON_PARAMS="12=1" python -m pytest tests/test_skeleton.py tests/test_golden.py tests/test_batch.py -x -q -m gpu 2>&1 | grep -E "passed|failed" >> $O
for P in "" "12=1" "12=1,6=1" "12=1,6=2" "12=1,6=8" "12=1,6=16" "12=1,6=8,7=16"; do
  echo "== ST_SKELETON_PARAMS=$P" >> $O
  ST_SKELETON_PARAMS=$P python tools/diag_phases.py 1000000 0.02 0 0 2>&1 | grep "^params" | cut -c1-150 >> $O
  ST_SKELETON_PARAMS=$P python tools/time_single.py 1000000 0.02 0 10 2>&1 | grep "ms per cloud" | cut -c1-100 >> $O
  for K in 20 384; do
  ST_SKELETON_PARAMS=$P ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps $K --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  steps $K: ms_per_step', round(d['ms_per_step'],3), 'skeleton_kernels', d['stage_ms'].get('skeleton_kernels'))" >> $O
  done
done
