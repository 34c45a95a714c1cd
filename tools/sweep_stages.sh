# GPU box: what the skeleton stage costs the batched pipeline, and the SSSP launch shape (ST_SKELETON_PARAMS 10 = workgroups,
# 8 = lanes per frontier vertex, 6 = levels per launch).  ST_DIAG_STAGES=1 runs SSSP only (no branch selection: zero branches).
export ST_BENCH_MIN_UPTIME_S=${ST_BENCH_MIN_UPTIME_S:-0}  # developer sweeps: no minimum warm-up time
cd $GRAFT_REPO_ROOT
run() { # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d.get('stage_ms',{}); print('$label: %.3f ms/cloud  skeleton_kernels %.3f' % (d['ms_per_step'], s.get('skeleton_kernels',0)))"
}
run "default" ST_X=0
run "default (repeat)" ST_X=0
run "SSSP only (no select)" ST_DIAG_STAGES=1
run "sssp blocks 512" ST_SKELETON_PARAMS=10=512
run "sssp blocks 1024" ST_SKELETON_PARAMS=10=1024
run "sssp blocks 2048" ST_SKELETON_PARAMS=10=2048
run "sssp blocks 1024, 16 lanes" ST_SKELETON_PARAMS=10=1024,8=16
run "sssp blocks 2048, 16 lanes" ST_SKELETON_PARAMS=10=2048,8=16
run "sssp blocks 1024, 32 lanes" ST_SKELETON_PARAMS=10=1024,8=32
run "sssp blocks 1024, 8 levels" ST_SKELETON_PARAMS=10=1024,6=8
run "default (end)" ST_X=0
