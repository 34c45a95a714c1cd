cd $GRAFT_REPO_ROOT
python tools/diag_phases.py 1000000 0.02 0 0 2>&1 | grep "^params" | cut -c1-120
python tools/diag_phases.py 5000000 0.01 0.6 3 2>&1 | grep "^params" | cut -c1-120
python tools/diag_phases.py 5000000 0.01 0.6 3 "10=256" 2>&1 | grep "^params" | cut -c1-120
python tools/diag_phases.py 5000000 0.01 0.6 3 "10=4096,8=32" 2>&1 | grep "^params" | cut -c1-120
python -m pytest tests/test_skeleton.py tests/test_batch.py tests/test_golden.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: %.3f ms/cloud, single %.2f ms, stages %s' % (d['ms_per_step'], d['config']['single_cloud_latency_ms'], d['stage_ms']))"; done
