# kernel trace of the multi-stream bench (default streams): is the GPU ever idle?  -> tools/trace_cover.py
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_s8
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_s8 -- python $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 8 --no-cpu-baseline ${ST_STREAMS:+--streams $ST_STREAMS} > $GRAFT_REPO_ROOT/gpurun_out/prof_s8.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_s8.log | cut -c1-400
python tools/trace_cover.py
