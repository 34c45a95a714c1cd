# GPU box: average duration of the SSSP / select launches of one batch of 24 clouds at a time under different launch shapes
export ST_BENCH_MIN_UPTIME_S=${ST_BENCH_MIN_UPTIME_S:-0}  # developer sweeps: no minimum warm-up time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in "6=4" "6=6" "6=8" "6=12" "6=8,8=32"; do
  rm -rf /tmp/prof_x
  ST_SKELETON_PARAMS=$p timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/bench.py --streams 1 --steps 48 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > /tmp/prof_x.log 2>&1
  f=$(ls /tmp/prof_x/*/*kernel_stats.csv | head -1)
  echo "== $p: $(grep -o '"ms_per_step": [0-9.]*' /tmp/prof_x.log | head -1)"
  python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if r['Name'].startswith(('k_sk_sssp_round','k_sk_select','k_sk_preds','k_sk_claim')): print('   %-18s calls %5s total %8.2f ms avg %8.1f us' % (r['Name'][:16], r['Calls'], int(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))"
done
