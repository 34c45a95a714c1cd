# GPU box: everything profiles/ is built from, in one call (then: python tools/summarize_profiles.py r01)
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh ${1:-r01}
cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_conv.py > gpurun_out/conv_1m.txt 2>&1
timeout 400 python tools/bench_conv.py 5000000 0.01 0.6 > gpurun_out/conv_5m.txt 2>&1
timeout 300 python tools/time_fp16.py > gpurun_out/fp16_1m.txt 2>&1
timeout 400 python tools/time_fp16.py 5000000 0.01 0.6 > gpurun_out/fp16_5m.txt 2>&1
bash tools/prof_solo.sh
tail -n 3 gpurun_out/fp16_1m.txt; tail -n 3 gpurun_out/fp16_5m.txt
