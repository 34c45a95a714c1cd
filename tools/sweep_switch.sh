cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3; do
  for cfg in ":8" "0.0001:8" "0.00002:8" "0.05:8" ":6"; do
    SW=${cfg%%:*}; S=${cfg##*:}
    echo -n "round $round switch=${SW:-default} S=$S: "
    ST_BENCH_SWITCH=$SW timeout 200 python bench.py --steps 48 --warmup 4 --streams $S --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
