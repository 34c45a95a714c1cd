# A/B on one box: enqueue-only entry points bound through PyDLL (GIL kept) vs CDLL (GIL dropped around every call)
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3; do
  for cfg in "0:4" "1:4" "0:16" "1:16"; do
    C=${cfg%%:*}; Q=${cfg##*:}
    echo -n "round $round cdll_only=$C Q=$Q S=8: "
    ST_CDLL_ONLY=$C GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --steps 48 --warmup 4 --streams 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
