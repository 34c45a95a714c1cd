# GPU box: the whole gpu-marked suite, then the default bench line
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests.txt 2>&1; tail -4 gpurun_out/r2_tests.txt
timeout 600 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; tail -3 gpurun_out/r2_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench.json').read())
print('value %.1f Mpts/s  %.3f ms/cloud  upload-incl %.1f  parity_in_run %s  cpu %.0f pts/s' % (d['value']/1e6, d['ms_per_step'], d['value_incl_host_upload']/1e6, d.get('parity_in_run'), d['cpu_baseline']['value']))
print('roofline', {k:d['roofline'][k] for k in ('kernel','bound','achieved','frac','avg_us','launches')})
print('conv in-pipeline %.1f%%  solo %.1f%%' % (100*d['roofline']['gather_gemm']['hbm_frac'], 100*d['roofline_solo']['gather_gemm']['hbm_frac']))
print(d['stage_ms']); print(d['config']); print(d.get('other_configs'))
PY
