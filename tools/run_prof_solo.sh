# GPU box: kernel statistics of one batch of 24 clouds at a time (solo durations), top kernels in us per cloud
export ST_BENCH_MIN_UPTIME_S=${ST_BENCH_MIN_UPTIME_S:-0}  # developer sweeps: no minimum warm-up time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python $R/bench.py --streams 1 --steps 96 --warmup 8 --batch 24 --no-cpu-baseline --no-extras > /tmp/prof_x.log 2>&1
f=$(ls /tmp/prof_x/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/s2_solo_kernel_stats.csv
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_x.log | head -1
python - <<P
import csv
rows=list(csv.DictReader(open('$f')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
clouds=24+2+96+96+24
print('total %.1f ms over %d clouds = %.3f ms/cloud'%(tot/1e6,clouds,tot/1e6/clouds))
for r in rows[:45]:
    print('%8.1f us/cloud %6s calls avg %8.1f us  %s'%(int(r['TotalDurationNs'])/1e3/clouds, r['Calls'], float(r['AverageNs'])/1e3, r['Name'][:110]))
P
