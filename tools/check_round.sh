# GPU box: parity tests of the touched stages, then the solo kernel profile and the bench line
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/check_tests.txt 2>&1; tail -3 gpurun_out/check_tests.txt
bash tools/prof_solo.sh
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; cut -c1-900 gpurun_out/check_bench.json
