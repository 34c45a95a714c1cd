# GPU box: does a bench that starts right after another GPU process (the round-end sequence: pytest -m gpu, smoke, bench) run slow?
cd $GRAFT_REPO_ROOT
b() { # label, env...
  label=$1; shift
  env "$@" timeout 400 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$label: %.3f ms/cloud (upload-inclusive pass %.3f) warm-up clouds %d  outlier_removal %.3f post_process %.3f  single cloud %.2f ms' % (d['ms_per_step'], 1e9/d['value_incl_host_upload'], d['config']['warmup_steps_run'], s['outlier_removal'], s['post_process'], d['config']['single_cloud_latency_ms']))"
}
for mode in "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0"; do
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b "right after pytest + smoke, $mode, warm-up >= 30 s of process life" $mode
done
