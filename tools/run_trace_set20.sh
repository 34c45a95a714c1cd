#!/bin/bash
# kernel trace of the driver's launch set (20 clouds, one set) folded in launch order (tools/trace_one_batch.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tb20
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tb20 -- python $R/bench.py --streams 1 --steps 20 --warmup 20 --no-cpu-baseline --no-extras > /tmp/prof_tb20.log 2>&1
for w in $(seq 1 30); do python $R/tools/trace_one_batch.py /tmp/prof_tb20 $w > /tmp/tb20_$w.txt 2>&1; head -1 /tmp/tb20_$w.txt; done > $R/gpurun_out/trace_set20_index.txt
# the last launch set of 20 clouds that ran with the timers off (wall 25-32 ms)
for w in $(seq 1 30); do if grep -q "wall \(2[5-9]\|3[0-2]\)\." /tmp/tb20_$w.txt; then cp /tmp/tb20_$w.txt $R/gpurun_out/trace_set20.txt; break; fi; done
# ... and one single-cloud call (process_cloud from pinned host memory: ~410 kernels)
for w in $(seq 1 30); do if grep -q "launch set: \(3[5-9]\|4[0-3]\)[0-9] kernels, wall \(9\|10\|11\)\." /tmp/tb20_$w.txt; then cp /tmp/tb20_$w.txt $R/gpurun_out/trace_single_cloud.txt; break; fi; done
