#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the bench line of the default command and of the driver's command (--steps 20), the
# rocprofv3 kernel statistics of both, the one-batch-at-a-time statistics, and the two PMC passes `roofline.traffic` comes from.
#   gpurun --timeout 2400 -- 'bash tools/collect_r3.sh r03'   then   python tools/summarize_profiles.py r03
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -s KILL 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 300 gpurun_out/${tag}_bench.json; echo
timeout -s KILL 600 python bench.py --steps 20 > gpurun_out/${tag}_bench_steps20.json 2> gpurun_out/${tag}_bench_steps20.err
tail -c 300 gpurun_out/${tag}_bench_steps20.json; echo
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}.log 2>&1
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_steps20 -- python $R/bench.py --steps 20 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}_steps20.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_solo.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
cd $R
# the per-dispatch traces are large (the merge back is capped at 64 MiB): keep the statistics and the PMC tables only
rm -f gpurun_out/prof_${tag}/*/*kernel_trace.csv gpurun_out/prof_${tag}_steps20/*/*kernel_trace.csv gpurun_out/prof_solo/*/*kernel_trace.csv gpurun_out/pmc_*/*/*kernel_trace.csv
python tools/time_single.py > gpurun_out/${tag}_single.txt 2>&1
ls gpurun_out/prof_${tag}/*/ gpurun_out/prof_${tag}_steps20/*/ gpurun_out/prof_solo/*/ gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/ 2>&1 | tail -20
# the multi-rank control flow on ONE GPU (two gloo ranks, both on cuda:0): sharding, the size all_gather + payload gather, all_reduce, barrier
ST_BENCH_DRYRUN=1 ST_BENCH_MIN_UPTIME_S=5 timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 16 --warmup 2 > gpurun_out/${tag}_dryrun_2ranks.json 2> gpurun_out/${tag}_dryrun_2ranks.err
tail -c 200 gpurun_out/${tag}_dryrun_2ranks.json; echo
timeout -s KILL 600 python tools/parity_stress.py 300 > gpurun_out/${tag}_parity_stress.txt 2>&1
tail -2 gpurun_out/${tag}_parity_stress.txt
