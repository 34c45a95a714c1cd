#!/bin/bash
# round 3: rounds per select launch (2) x small_work (1) x local_items (4): the driver's --steps 20 and the 384-step default
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_select2.txt; : > $O
for P in "" "2=100000,4=1073741824" "2=100000,4=1073741824,1=1048576" "2=100000,4=1073741824,1=16777216" "2=100000,4=4000000" ; do
  echo "== ST_SKELETON_PARAMS=$P" >> $O
  for K in 20 384; do
  ST_SKELETON_PARAMS=$P ST_BENCH_MIN_UPTIME_S=12 python bench.py --steps $K --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  steps $K: ms_per_step', round(d['ms_per_step'],3), 'last warm', d['config']['last_warmup_pass_ms_per_step'], 'single', d['config']['single_cloud_latency_ms'], 'stage skeleton_kernels', d['stage_ms'].get('skeleton_kernels'))" >> $O
  done
done
