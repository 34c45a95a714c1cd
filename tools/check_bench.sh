# bench.py in its modes: default, short runs, --procs, the 2-rank dry run on one GPU
cd $GRAFT_REPO_ROOT
python -c "import torch; torch.zeros(1).cuda()" 2>/dev/null
run() { echo "== $*"; timeout 400 "$@" 2>gpurun_out/cb.err | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus', 'steps')}, d['config']['clouds_in_flight_per_gpu'], d['config'].get('worker_processes_per_gpu'), d['config']['single_stream_ms_per_cloud'], d['roofline']['kernel'], round(d['roofline']['avg_us'], 1), d['last_result'], 'cpu' in str(d.get('cpu_baseline', {}).get('kind', '')) or d.get('cpu_baseline', {}).get('kind'))
except Exception as e:
    print('FAILED', e, l[:300]); print(open('gpurun_out/cb.err').read()[-1500:])
"; }
run python bench.py --no-cpu-baseline
run python bench.py --steps 1 --warmup 0 --no-cpu-baseline
run python bench.py --steps 5 --warmup 1 --no-cpu-baseline
run python bench.py --steps 48 --warmup 4 --procs 2 --streams 4 --no-cpu-baseline
run python bench.py --steps 48 --warmup 4 --procs 3 --streams 4 --no-cpu-baseline
run python bench.py --steps 48 --warmup 4 --no-cpu-baseline
ST_BENCH_DRYRUN=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 12 --warmup 2
ST_BENCH_DRYRUN=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 12 --warmup 2 --procs 2 --streams 2
