# GPU box: the whole -m gpu suite, the per-layer conv tables (configs[1] and configs[3] sizes), the solo kernel profile
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/check_tests.txt 2>&1; tail -3 gpurun_out/check_tests.txt
timeout 300 python tools/bench_conv.py > gpurun_out/conv_1m.txt 2>&1; grep " up \| down " gpurun_out/conv_1m.txt | cut -c1-200
timeout 400 python tools/bench_conv.py 5000000 0.01 0.6 > gpurun_out/conv_5m.txt 2>&1; grep " up \| down " gpurun_out/conv_5m.txt | cut -c1-200
bash tools/prof_solo.sh
