# hardware queues x clouds in flight (HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default)
cd $GRAFT_REPO_ROOT
for Q in 4 8 16; do for S in 8 12 16; do
  echo -n "GPU_MAX_HW_QUEUES=$Q streams=$S: "
  GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --steps 48 --warmup 4 --streams $S --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done
