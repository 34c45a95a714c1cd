# GPU box: clouds per launch set x batches in flight
export ST_BENCH_MIN_UPTIME_S=${ST_BENCH_MIN_UPTIME_S:-0}  # developer sweeps: no minimum warm-up time
cd $GRAFT_REPO_ROOT
for cfg in ${SWEEP:-"2 8" "3 8" "4 8" "2 16" "3 16" "4 16" "2 32"}; do set -- $cfg
  timeout 300 python bench.py --streams $1 --batch $2 --steps ${STEPS:-96} --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 batch $2: %.3f ms/cloud  %.1f Mpts/s  incl upload %.1f  conv agg %.1f%%' % (d['ms_per_step'], d['value']/1e6, d['value_incl_host_upload']/1e6, 100*d['roofline']['gather_gemm']['hbm_frac']), d['stage_ms'])"
done
