cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/parity_stress.py 2>&1 | tail -1
python tools/diag_phases.py 1000000 0.02 0 0 2>&1 | grep "^params\|phases" | cut -c1-330
python tools/diag_phases.py 5000000 0.01 0.6 3 2>&1 | grep "^params\|phases" | cut -c1-330
bash tools/run_prof_solo.sh 2>&1 | grep "ms_per_step\|total\|k_sk_select\|k_sk_sssp\|k_sk_claim"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: %.3f ms/cloud, single %.2f ms, frac %.3f' % (d['ms_per_step'], d['config']['single_cloud_latency_ms'], d['roofline']['frac']))"; done
