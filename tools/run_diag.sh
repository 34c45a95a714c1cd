cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/parity_stress.py 2>&1 | tail -3
bash tools/run_prof_solo.sh 2>&1 | grep "ms_per_step\|total\|k_vx_pass\|k_sk_preds\|k_sk_sssp"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench: %.3f ms/cloud' % (d['ms_per_step']))"; done
