cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/check_tests.txt 2>&1; tail -3 gpurun_out/check_tests.txt
bash tools/prof_solo.sh
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 300 python bench.py --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_stream_ms_per_cloud": [0-9.]*' | tr '\n' ' '; echo; done
