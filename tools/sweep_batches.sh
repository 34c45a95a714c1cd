# alternating A/B runs on ONE box: knobs of the progress read-backs in the SSSP (6: levels per launch, 9: first batch in
# units of 32 launches) and select (3: first batch of launch pairs) loops; 8 clouds in flight and one cloud at a time
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3 4; do
  for cfg in ${CFGS:-"6=4" "6=6" "6=8" "6=4,9=3" "6=8,9=1"}; do
    echo -n "round $round params=$cfg: "
    ST_SKELETON_PARAMS=$cfg timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_stream_ms_per_cloud": [0-9.]*' | tr '\n' ' '; echo
  done
done
