#!/bin/bash
# round 3: the driver's --steps 20, free-running, two repeats per plan
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_plan3.txt; : > $O
for rep in 1 2; do for PS in "10,10 2 3" "10,10 2 0" "12,8 2 0" "13,7 2 0" "11,9 2 0" "14,6 2 0" "11,6,3 3 0" "9,7,4 3 0" "8,7,5 3 0"; do
  set -- $PS
  ST_BENCH_PLAN=$1 ST_BENCH_ORDERED=$3 ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps 20 --warmup 3 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('plan $1 streams $2 ordered $3: ms_per_step', round(d['ms_per_step'],3), 'last warm', c['last_warmup_pass_ms_per_step'])" >> $O
done; done
