#!/bin/bash
# round 3: the driver's --steps 20 under different batch plans / schedules
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_plan.txt; : > $O
for ORD in 3 0; do for BS in "64 2" "10 3" "7 3" "5 3" "4 3" "20 1"; do
  set -- $BS
  echo "== ORDERED=$ORD --batch $1 --streams $2" >> $O
  ST_BENCH_ORDERED=$ORD ST_BENCH_MIN_UPTIME_S=10 python bench.py --steps 20 --warmup 3 --batch $1 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('  ms_per_step', round(d['ms_per_step'],3), 'last warm', c['last_warmup_pass_ms_per_step'], 'plan', c['clouds_per_launch_set'], 'x', c['batches_in_timed_region'], 'in flight', c['batches_in_flight_per_gpu'])" >> $O
done; done
