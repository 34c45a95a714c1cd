#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): everything profiles/r04_* is made of.
#   gpurun --timeout 2400 -- 'bash tools/collect_r4.sh r04'   then   python tools/summarize_profiles.py r04
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
# the driver's command and the default command: the JSON lines
timeout -s KILL 600 python bench.py --steps 20 > gpurun_out/${tag}_bench_steps20.json 2> gpurun_out/${tag}_bench_steps20.err
tail -c 200 gpurun_out/${tag}_bench_steps20.json; echo
timeout -s KILL 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 200 gpurun_out/${tag}_bench.json; echo
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel statistics of the same two commands, and one batch at a time (solo kernel durations)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_steps20 -- python $R/bench.py --steps 20 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}_steps20.log 2>&1
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_solo.log 2>&1
# HBM traffic: two separate PMC passes, one batch of 16 clouds per pass
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
cd $R
rm -f gpurun_out/prof_${tag}/*/*kernel_trace.csv gpurun_out/prof_${tag}_steps20/*/*kernel_trace.csv gpurun_out/prof_solo/*/*kernel_trace.csv gpurun_out/pmc_*/*/*kernel_trace.csv
# one cloud at a time; the branch selection's phases per seed; the SSSP level sweep; the grid barrier
python tools/time_single.py > gpurun_out/${tag}_single.txt 2>&1
( for seed in 0 1 2 3; do for p in "" "12=256"; do echo "== seed $seed params [$p]"; timeout 120 python tools/diag_phases.py 1000000 0.02 0 $seed "$p" 2>&1 | tail -2; done; done ) > gpurun_out/${tag}_select_by_seed.txt 2>&1
( for p in "" "6=1" "6=2" "6=3" "6=6" "6=8" "12=1" "12=1,6=1" "12=1,6=2" "12=1,6=8" "13=0" "8=64" "10=256"; do echo "== params [$p]"; timeout 120 python tools/diag_phases.py 1000000 0.02 0 0 "$p" 2>&1 | grep "SSSP + pred" | cut -c1-140; done ) > gpurun_out/${tag}_sssp_levels.txt 2>&1
( timeout 60 ./tools/microbench/grid_barrier ) > gpurun_out/${tag}_grid_barrier.txt 2>&1
( timeout 200 python tools/probe_voxel_order.py 2>&1 | grep "us per call" ) > gpurun_out/${tag}_voxel_order.txt 2>&1
( timeout 120 python tools/probe_h2d.py 2>&1 | grep "GB/s" ) > gpurun_out/${tag}_h2d.txt 2>&1
# the pass of the driver's command on a timeline
bash tools/run_trace_steps20.sh > /dev/null 2>&1
cp gpurun_out/trace_steps20.txt gpurun_out/${tag}_timeline_steps20.txt
# the multi-rank control flow on ONE GPU (two gloo ranks, both on cuda:0)
ST_BENCH_DRYRUN=1 ST_BENCH_MIN_UPTIME_S=5 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 2 > gpurun_out/${tag}_dryrun_2ranks.json 2> gpurun_out/${tag}_dryrun_2ranks.err
timeout -s KILL 600 python tools/parity_stress.py 300 > gpurun_out/${tag}_parity_stress.txt 2>&1
tail -2 gpurun_out/${tag}_parity_stress.txt
( for b in 10 12; do echo "== --steps 20 --batch $b"; timeout 300 python bench.py --steps 20 --batch $b --no-cpu-baseline --no-extras 2>/dev/null | python tools/show_bench.py /dev/stdin | head -3; done ) > gpurun_out/${tag}_steps20_plans.txt 2>&1
( timeout 400 python tools/probe_canopy.py "" "12=256" 2>&1 | grep "^params" ) > gpurun_out/${tag}_canopy_probe_final.txt 2>&1
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/${tag}_gpu_tests.txt 2>&1
cat gpurun_out/${tag}_gpu_tests.txt
