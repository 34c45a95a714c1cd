# does the number of hardware queues bound the 8-in-flight rate?  alternating rounds on ONE box
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3; do
  for q in 1 2 4 8; do
    echo -n "round $round GPU_MAX_HW_QUEUES=$q: "
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_stream_ms_per_cloud": [0-9.]*' | tr '\n' ' '; echo
  done
done
