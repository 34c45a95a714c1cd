#!/bin/bash
# round 3, second session: parity tests of the touched stages, the driver's 20 steps (and other batch plans), the default run, one
# cloud alone, and the folded kernel trace of one 24-cloud launch set
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_check_${1:-x}.txt; : > $O
cd $R
python -m pytest tests/test_skeleton.py tests/test_golden.py tests/test_batch.py tests/test_voxelize.py tests/test_whole_cloud_mode.py -x -q -m gpu 2>&1 | tail -3 >> $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms_per_step', round(d['ms_per_step'],3), 'single', d['config'].get('single_cloud_latency_ms'), 'stage_ms', {k: round(v,3) for k,v in d.get('stage_ms',{}).items()})"; }
for K in 20 384; do
ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps $K --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line "steps $K" >> $O 2>&1
done
for P in $2; do
ST_BENCH_PLAN=$P ST_BENCH_MIN_UPTIME_S=8 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | line "steps 20 plan $P" >> $O 2>&1
done
python tools/time_single.py 1000000 0.02 0 10 2>&1 | grep -E "ms per cloud|stage brackets" | cut -c1-400 >> $O
bash tools/run_trace_batch.sh
