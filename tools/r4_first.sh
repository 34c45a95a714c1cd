#!/bin/bash
# round 4, first GPU session: parity of the new branch selection, its phases, one cloud at a time, the driver's command
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_skeleton.py tests/test_batch.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r4_tests.txt 2>&1
( for seed in 0 1; do for p in "" "14=1,1=1048576,2=1048576,3=1"; do echo "== seed $seed params [$p]"; timeout 300 python tools/diag_phases.py 1000000 0.02 0 $seed "$p" 2>&1 | tail -4; done; done ) > gpurun_out/r4_phases.txt 2>&1
( timeout 300 python tools/time_single.py 2>&1 | tail -3 ) > gpurun_out/r4_single.txt 2>&1
( timeout 120 python tools/probe_h2d.py 2>&1 | tail -14 ) > gpurun_out/r4_h2d.txt 2>&1
( timeout 600 python bench.py --steps 20 --no-extras 2>&1 | tail -1 ) > gpurun_out/r4_bench20.json 2>&1
