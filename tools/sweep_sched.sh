cd $GRAFT_REPO_ROOT
for SCHED in "" spin yield block; do for S in 8 12; do
  echo -n "sched=${SCHED:-default} Q=16 streams=$S: "
  ST_BENCH_SCHED=$SCHED GPU_MAX_HW_QUEUES=16 timeout 200 python bench.py --steps 48 --warmup 4 --streams $S --no-cpu-baseline 2>/tmp/err.txt | grep -o '"ms_per_step": [0-9.]*'
  grep hipSetDeviceFlags /tmp/err.txt
done; done
