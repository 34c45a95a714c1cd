#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the official bench line, the rocprofv3 kernel statistics of the same command
# and the two PMC passes the roofline's `traffic` comes from.  Every step has a hard timeout: rocprofv3 has been seen
# to hang after its child exits with an error.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r01'   then   python tools/summarize_profiles.py r01
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout -s KILL 420 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${tag}.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R; ls gpurun_out/prof_${tag}/*/ gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/ 2>&1 | tail -12
