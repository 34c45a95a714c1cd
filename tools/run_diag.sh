cd $GRAFT_REPO_ROOT
p() { python -c "
import json,sys
d=json.load(open('$1')); s=d['stage_ms']; print('$2: %.3f ms/cloud, upload-inclusive %.3f, outlier_removal stage %.3f, warm-up clouds %d' % (d['ms_per_step'], 1e9/d['value_incl_host_upload'], s['outlier_removal'], d['config']['warmup_steps_run']))"; }
timeout -s KILL 600 python bench.py > /tmp/a.json 2>/dev/null; p /tmp/a.json "full (1)"
timeout -s KILL 600 python bench.py --no-cpu-baseline --no-extras > /tmp/b.json 2>/dev/null; p /tmp/b.json "no baseline / extras (2)"
timeout -s KILL 600 python bench.py > /tmp/c.json 2>/dev/null; p /tmp/c.json "full (3)"
timeout -s KILL 600 python bench.py --no-extras > /tmp/d.json 2>/dev/null; p /tmp/d.json "no extras (4)"
timeout -s KILL 600 python bench.py --no-cpu-baseline > /tmp/e.json 2>/dev/null; p /tmp/e.json "no baseline (5)"
nproc; uptime
