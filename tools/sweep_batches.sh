# alternating A/B runs on ONE box: launches per progress read-back in the SSSP (knob 9: first batch, in units of 32) and
# select (knob 3: first batch) loops, 8 clouds in flight and one cloud at a time
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3; do
  for cfg in "3=16,9=2" "3=24,9=3" "3=16,9=3" "3=24,9=2" "3=32,9=3"; do
    echo -n "round $round params=$cfg: "
    ST_SKELETON_PARAMS=$cfg timeout 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_stream_ms_per_cloud": [0-9.]*' | tr '\n' ' '; echo
  done
done
