cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3 4; do
  for Q in 4 8; do
    echo -n "round $round GPU_MAX_HW_QUEUES=$Q S=8: "
    GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --steps 48 --warmup 4 --streams 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
