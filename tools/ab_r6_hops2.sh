cd $GRAFT_REPO_ROOT
for p in "6=6" "6=5" "6=4" "6=8"; do
  python tools/probe_gt.py 20 "$p" 0 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
one=[v for k,v in d.items() if isinstance(v,dict) and 'one cloud per call' in k][0]
st=[v for k,v in d.items() if isinstance(v,dict) and 'one launch set' in k][0]
print('GT params [%s]: one cloud %.2f ms (%d SSSP launches); set of 20: %.2f ms (skeleton kernels %.3f per cloud)' % (d['params'], one['ms'], one['sssp_rounds'], st['ms_per_set'], st['stage_ms_per_cloud']['skeleton_kernels']))"
done
for rep in 1 2 3; do for p in "6=4" "6=6" "6=8"; do
  ST_SKELETON_PARAMS="$p" ST_BENCH_MIN_UPTIME_S=10 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('net [$p] %.1f M points/s, %.4f ms/step; skeleton kernels %.4f ms/cloud in the set; single cloud %.2f ms (skeleton kernels %.2f)' % (d['value']/1e6, d['ms_per_step'], d['stage_ms']['skeleton_kernels'], d['single_cloud']['ms'], d['single_cloud']['stage_ms']['skeleton_kernels']))"
done; done
