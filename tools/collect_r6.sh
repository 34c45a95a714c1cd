#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): everything profiles/r06_* is made of.
#   gpurun --timeout 2700 -- 'bash tools/collect_r6.sh r06'   then   python tools/summarize_profiles.py r06
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
# the driver's command and the default command: the JSON lines
timeout -s KILL 700 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_steps20.json 2> gpurun_out/${tag}_bench_steps20.err
tail -c 200 gpurun_out/${tag}_bench_steps20.json; echo
timeout -s KILL 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 200 gpurun_out/${tag}_bench.json; echo
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel statistics of the same two commands, and one batch at a time (solo kernel durations)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_steps20 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}_steps20.log 2>&1
timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}.log 2>&1
rm -rf $R/gpurun_out/prof_solo
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_solo.log 2>&1
# HBM traffic: two separate PMC passes, one batch of 16 clouds per pass
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
cd $R
rm -f gpurun_out/prof_${tag}/*/*kernel_trace.csv gpurun_out/prof_${tag}_steps20/*/*kernel_trace.csv gpurun_out/prof_solo/*/*kernel_trace.csv gpurun_out/pmc_*/*/*kernel_trace.csv
( timeout 700 bash tools/pmc_mfma.sh ) > gpurun_out/${tag}_pmc_mfma.txt 2>&1
# one cloud at a time; the branch selection's phases on the network's graph (seeds 0-3) and on the ground-truth graph
timeout 300 python tools/time_single.py > gpurun_out/${tag}_single.txt 2>&1
( for seed in 0 1 2 3; do echo "== seed $seed"; timeout 120 python tools/diag_phases.py 1000000 0.02 0 $seed "" 2>&1 | tail -2; done ) > gpurun_out/${tag}_select_by_seed.txt 2>&1
( timeout 300 python tools/probe_gt.py 20 "" 1 1 2>&1 | tail -1 ) > gpurun_out/${tag}_ground_truth_probe.json 2>&1
( for p in "" "6=1" "6=16"; do echo "== params [$p]"; timeout 200 python tools/probe_gt.py 1 "$p" 0 0 2>&1 | tail -1 | python tools/pp_probe.py; done ) > gpurun_out/${tag}_sssp_levels_ground_truth.txt 2>&1
# the multi-rank control flow and the host side of eight ranks on ONE GPU (gloo, every rank on cuda:0): NOT a scaling measurement
ST_BENCH_DRYRUN=1 timeout -s KILL 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 > gpurun_out/${tag}_dryrun_8ranks_steps20.json 2> gpurun_out/${tag}_dryrun_8ranks_steps20.err
# BASELINE.json configs[2] as written: a FIXED batch of 64 clouds split over 8 ranks (strong scaling) -- control flow only, one GPU under all ranks
ST_BENCH_DRYRUN=1 timeout -s KILL 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 8 --scaling strong --clouds 64 --warmup 2 > gpurun_out/${tag}_dryrun_8ranks_strong64.json 2> gpurun_out/${tag}_dryrun_8ranks_strong64.err
timeout -s KILL 700 python tools/parity_stress.py 300 > gpurun_out/${tag}_parity_stress.txt 2>&1
tail -2 gpurun_out/${tag}_parity_stress.txt
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.txt 2>&1
tail -3 gpurun_out/${tag}_gpu_tests.txt
