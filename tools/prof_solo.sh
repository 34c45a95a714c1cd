# single-stream kernel statistics (solo kernel durations, no overlap between clouds)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_solo -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_solo.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_solo.log | cut -c1-300
