# GPU box: HBM traffic of the skeleton kernels (two PMC passes), one batch at a time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --streams 1 --steps 16 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_write.log 2>&1
cd $R; python - <<'PY'
import csv,glob,collections
def counter(pattern,name):
    f=sorted(glob.glob(pattern))[-1]; agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==name: agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return agg
fe=counter("gpurun_out/pmc_fetch/*/*counter_collection.csv","FETCH_SIZE"); wr=counter("gpurun_out/pmc_write/*/*counter_collection.csv","WRITE_SIZE")
for k in ("k_sk_select","k_sk_claim","k_sk_sssp_round"):
    if k in fe: print(k, "launches", len(fe[k]), "fetch KB/launch %.1f" % (sum(fe[k])/len(fe[k])), "write KB/launch %.1f" % (sum(wr[k])/len(wr[k])))
PY
