# A/B of two revisions on ONE box: `git archive <rev> | tar -x -C _ab_old`, build it there (python -c "import __graft_entry__ as g; g.build()"), then gpurun this script
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
for round in 1 2 3 4 5 6; do
  for d in _ab_old .; do
    echo -n "round $round rev=$([ $d = . ] && echo HEAD || echo 81ea9bd): "
    (cd $GRAFT_REPO_ROOT/$d && timeout 200 python bench.py --steps 96 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_stream_ms_per_cloud": [0-9.]*' | tr '\n' ' '); echo
  done
done
