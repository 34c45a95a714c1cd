# GPU box: kernel statistics of the batched path, one batch at a time on one stream (solo kernel durations per batch of 8)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_batch1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --batch ${1:-8} --steps 32 --warmup 8 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_batch1.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_batch1.log | cut -c1-400
f=$(ls -t gpurun_out/prof_batch1/*/*kernel_stats.csv | head -1); head -60 $f | cut -c1-150
