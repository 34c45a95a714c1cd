#!/bin/bash
# round 3: SSSP frontier launch knobs at 10 clouds per launch set (the driver's --steps 20): workgroups (10), levels per launch (6), lanes per vertex (8), launches per read-back (7)
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_sssp.txt; : > $O
for P in "" "10=256" "10=512" "10=1024" "6=8" "6=8,10=512" "6=2" "8=16" "8=32" "6=8,7=16" "13=0"; do
  ST_SKELETON_PARAMS=$P ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$P] ms_per_step', round(d['ms_per_step'],3), 'skeleton_kernels', d['stage_ms'].get('skeleton_kernels'))" >> $O
done
