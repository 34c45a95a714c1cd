# GPU box: memory-pipeline counters of the sparse-conv kernels at the launch-set size (twenty clouds), one rocprofv3 --pmc pass per
# counter group (tools/bench_conv.py 1000000 0.02 0 20 1 2: every layer, split-bf16 kernel with two row tiles per wavefront).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VmemLatency MemUnitStalled" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM" "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TA_TA_BUSY TCP_TCC_READ_REQ_LATENCY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i + 1))
  rm -rf /tmp/pmcc_$i
  timeout -s KILL 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "sparse_conv" --output-format csv -d /tmp/pmcc_$i -- python $R/tools/bench_conv.py 1000000 0.02 0 20 1 2 > /tmp/pmcc_$i.log 2>&1
  echo "pass $i ($grp): rc $?"
done
python - <<P
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, 6):
    for f in glob.glob('/tmp/pmcc_%d/*/*counter_collection.csv' % i):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = ['VmemLatency', 'MemUnitStalled', 'TCP_TOTAL_CACHE_ACCESSES', 'TCP_TCC_READ_REQ', 'TCP_UTCL1_TRANSLATION_MISS', 'TCP_UTCL1_TRANSLATION_HIT', 'SQ_WAVE_CYCLES',
         'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VMEM', 'TCP_PENDING_STALL_CYCLES', 'TCP_TCP_TA_DATA_STALL_CYCLES', 'TA_TA_BUSY', 'TCP_TCC_READ_REQ_LATENCY',
         'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_WAVES']
print('kernel, dispatches | ' + ' | '.join(names) + '   (mean per dispatch)')
for k in sorted(agg):
    if 'b3' in k or 'k_sparse_conv<' in k:
        a = agg[k]
        print(k[:44].ljust(44), len(a.get(names[0], [])), ' | '.join('%.4g' % (sum(a[n]) / len(a[n])) if a.get(n) else '-' for n in names))
P
