cd $GRAFT_REPO_ROOT
for p in "1=4194304,2=16" "1=4194304,2=8" "1=2097152,2=8" "1=1048576,2=8" "1=4194304,2=4" "1=4194304,2=8,3=32" "1=4194304,2=12"; do
  ST_SKELETON_PARAMS=$p timeout 300 python bench.py --streams 1 --batch 16 --steps 64 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels'].get('k_sk_select',{}); print('params [$p]: %.3f ms/cloud  skeleton %.3f  select total %.1f ms in %d launches' % (d['ms_per_step'], d['stage_ms']['skeleton_kernels'], r.get('total_ms',0), r.get('launches',0)))"
done
