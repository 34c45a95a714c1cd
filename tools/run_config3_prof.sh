# GPU box: kernel statistics of configs[3] (one 5M-point cloud at a time)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python $R/tools/time_config3.py > $R/gpurun_out/prof_c3.log 2>&1
cd $R; f=$(ls gpurun_out/prof_c3/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-160
