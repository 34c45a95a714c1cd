#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the official bench line, the rocprofv3 kernel statistics of the same command, the
# one-batch-at-a-time statistics, and the two PMC passes the roofline's `traffic` comes from.  Hard timeouts everywhere.
#   gpurun --timeout 1500 -- 'bash tools/collect_r2.sh r02'   then   python tools/summarize_profiles.py r02
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -s KILL 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 400 gpurun_out/${tag}_bench.json
# for the record: the free-running schedule beside the ordered one (alternating, same box)
for i in 1 2; do
  ST_BENCH_MIN_UPTIME_S=12 timeout -s KILL 300 python bench.py --free-running --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['free_running']; r=d['roofline']
print('ordered (default, %d batches in flight): %.3f ms per cloud = %.1f M points/s, sparse-conv family at %.1f %% of the HBM peak in the timed region | free-running (%d in flight): %.3f ms per cloud = %.1f M points/s, %.1f %%' % (d['config']['batches_in_flight_per_gpu'], d['ms_per_step'], d['value']/1e6, 100*r['frac'], f['batches_in_flight'], f['ms_per_step'], f['value']/1e6, 100*f['roofline_frac']))"
done > gpurun_out/${tag}_free_running.txt
cat gpurun_out/${tag}_free_running.txt
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${tag}.log 2>&1
ST_BENCH_MIN_UPTIME_S=0 timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_solo -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_solo.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
cd $R; ls gpurun_out/prof_${tag}/*/ gpurun_out/prof_solo/*/ gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/ 2>&1 | tail -16
