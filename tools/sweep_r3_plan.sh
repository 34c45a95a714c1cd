#!/bin/bash
# round 3: the driver's --steps 20 under explicit batch plans (tapered: the last batches are the ones whose skeleton stage is exposed)
O=$GRAFT_REPO_ROOT/gpurun_out/r3_sweep_plan2.txt; : > $O
for ORD in 3 0; do for PS in "10,10 2" "12,8 2" "8,6,4,2 2" "8,6,4,2 3" "7,6,4,3 3" "10,6,4 3" "9,6,3,2 3" "6,5,4,3,2 3" "14,6 2" "11,6,3 3" "8,5,4,2,1 3"; do
  set -- $PS
  echo "== ORDERED=$ORD plan $1 streams $2" >> $O
  ST_BENCH_PLAN=$1 ST_BENCH_ORDERED=$ORD ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps 20 --warmup 3 --streams $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('  ms_per_step', round(d['ms_per_step'],3), 'last warm', c['last_warmup_pass_ms_per_step'], 'in flight', c['batches_in_flight_per_gpu'])" >> $O
done; done
