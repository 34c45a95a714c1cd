# alternating A/B runs on ONE box (boxes differ by +-20 %): hardware queue count and host wait policy at 8 clouds in flight
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null   # page the image in before the first timed run
for round in 1 2 3; do
  for cfg in "4:" "8:" "16:" "16:block" "32:"; do
    Q=${cfg%%:*}; SCHED=${cfg##*:}
    echo -n "round $round Q=$Q sched=${SCHED:-default} S=8: "
    ST_BENCH_SCHED=$SCHED GPU_MAX_HW_QUEUES=$Q timeout 200 python bench.py --steps 48 --warmup 4 --streams 8 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
  done
done
