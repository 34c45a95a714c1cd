cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3 4; do
  for P in 0 1; do
    echo -n "round $round poll=$P S=8: "
    ST_SYNC_POLL=$P timeout 200 python bench.py --steps 48 --warmup 4 --streams 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
