#!/bin/bash
# GPU box: the round's LAST revision once more -- the two JSON lines, the kernel statistics of the driver's command, the ground-truth
# probe, the launch-set listings and the GPU suite (the PMC passes, dry runs and the parity stress of collect_r6.sh are not repeated)
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R
timeout -s KILL 700 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_steps20.json 2> gpurun_out/${tag}_bench_steps20.err
timeout -s KILL 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}_steps20
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_steps20 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}_steps20.log 2>&1
cd $R
rm -f gpurun_out/prof_${tag}_steps20/*/*kernel_trace.csv
( timeout 300 python tools/probe_gt.py 20 "" 1 1 2>&1 | tail -1 ) > gpurun_out/${tag}_ground_truth_probe.json 2>&1
timeout 300 python tools/time_single.py > gpurun_out/${tag}_single.txt 2>&1
bash tools/run_trace_set20.sh > gpurun_out/${tag}_trace.log 2>&1
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -3 gpurun_out/${tag}_gpu_tests.txt
