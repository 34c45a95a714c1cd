#!/bin/bash
# round 3: the long-path claim inside the workgroup (14) x rounds per select launch (2) x small_work (1)
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_select3.txt; : > $O
python -m pytest tests/test_skeleton.py tests/test_golden.py tests/test_batch.py -x -q -m gpu 2>&1 | tail -2 >> $O
for P in "14=0" "" "2=100000" "2=100000,1=1048576" "2=100000,1=4194304" "2=100000,3=4" "2=64"; do
  echo "== ST_SKELETON_PARAMS=$P" >> $O
  ST_SKELETON_PARAMS=$P python tools/time_single.py 1000000 0.02 0 10 2>&1 | grep "ms per cloud" | cut -c1-120 >> $O
  for K in 20 384; do
  ST_SKELETON_PARAMS=$P ST_BENCH_MIN_UPTIME_S=10 python bench.py --steps $K --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  steps $K: ms_per_step', round(d['ms_per_step'],3), 'last warm', d['config']['last_warmup_pass_ms_per_step'], 'stage skeleton_kernels', d['stage_ms'].get('skeleton_kernels'))" >> $O
  done
done
