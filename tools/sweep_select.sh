# GPU box: select-stage knobs of the batched path, one batch at a time on one stream (ST_SKELETON_PARAMS, csrc/skeleton.hip)
export ST_BENCH_MIN_UPTIME_S=${ST_BENCH_MIN_UPTIME_S:-0}  # developer sweeps: no minimum warm-up time
cd $GRAFT_REPO_ROOT
for p in "" "1=1048576" "1=4194304" "1=16777216" "1=4194304,5=4194304" "2=8" "2=16" "1=4194304,2=16" "4=32768" "4=131072" "6=8" "6=16" "8=16"; do
  ST_SKELETON_PARAMS=$p timeout 300 python bench.py --streams 1 --batch 8 --steps 32 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels'].get('k_sk_select',{}); print('params [$p]: %.3f ms/cloud  skeleton %.3f  select total %.1f ms in %d launches' % (d['ms_per_step'], d['stage_ms']['skeleton_kernels'], r.get('total_ms',0), r.get('launches',0)))"
done
