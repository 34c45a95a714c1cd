cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sel -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_sel.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_sel.log | cut -c1-400
f=$(ls -t gpurun_out/prof_sel/*/*kernel_stats.csv | head -1); head -40 $f | cut -c1-140
