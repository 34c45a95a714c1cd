cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "unet or full_size or pipeline or abi" > gpurun_out/check_tests.txt 2>&1; tail -3 gpurun_out/check_tests.txt
timeout 300 python tools/bench_conv.py > gpurun_out/conv_1m.txt 2>&1; grep " up " gpurun_out/conv_1m.txt | cut -c1-200
timeout 400 python tools/bench_conv.py 5000000 0.01 0.6 > gpurun_out/conv_5m.txt 2>&1; grep " up " gpurun_out/conv_5m.txt | cut -c1-200
bash tools/prof_solo.sh
