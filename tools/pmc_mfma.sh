# GPU box: matrix-core utilisation of the sparse-conv kernels (rocprofv3 derived counter MfmaUtil = sum SQ_VALU_MFMA_BUSY_CYCLES /
# (GRBM_GUI_ACTIVE x SIMDs); VALUBusy in a second pass), one batch of 16 clouds, as the FETCH / WRITE passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in MfmaUtil VALUBusy; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > /tmp/pmc_$c.log 2>&1
done
python - <<P
import csv, glob, collections
def load(c):
    agg = collections.defaultdict(list)
    for f in glob.glob('/tmp/pmc_%s/*/*counter_collection.csv' % c):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return agg
m, v = load('MfmaUtil'), load('VALUBusy')
print('kernel, launches, MfmaUtil %, VALUBusy %   (per-dispatch values averaged; 16 clouds per launch)')
for k in sorted(m, key=lambda k: -sum(m[k])):
    if 'sparse_conv' in k or 'k_knn' in k or 'k_vx_insert' in k or 'k_sk_select' in k:
        print('%-46s %4d  %6.1f  %6.1f' % (k[:46], len(m[k]), sum(m[k]) / len(m[k]), sum(v.get(k, [0])) / max(len(v.get(k, [0])), 1)))
P
