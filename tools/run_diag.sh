cd $GRAFT_REPO_ROOT
p() { python -c "
import json,sys
d=json.load(open('$1')); s=d['stage_ms']; print('$2: %.3f ms/cloud (last warm-up pass %.3f), upload-inclusive %.3f, outlier_removal stage %.3f' % (d['ms_per_step'], d['config']['last_warmup_pass_ms_per_step'], 1e9/d['value_incl_host_upload'], s['outlier_removal']))"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for i in 1 2 3 4 5 6; do
timeout -s KILL 600 python bench.py > /tmp/a.json 2>/dev/null; p /tmp/a.json "full default command ($i)"
done
uptime
