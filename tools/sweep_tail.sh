cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3 4; do
  for cfg in "24:8" "24:6" "24:4" "12:4" "12:3" "12:2"; do
    K=${cfg%%:*}; S=${cfg##*:}
    echo -n "round $round steps=$K S=$S: "
    timeout 200 python bench.py --steps $K --warmup 2 --streams $S --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
