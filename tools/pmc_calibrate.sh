# GPU box: FETCH_SIZE / WRITE_SIZE of kernels with known byte counts (tools/microbench/pmc_calibrate.hip), two passes
R=$GRAFT_REPO_ROOT
cd $R/tools/microbench && hipcc --offload-arch=gfx950 -O2 -o pmc_calibrate pmc_calibrate.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal_f /tmp/cal_w
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal_f -- $R/tools/microbench/pmc_calibrate > /tmp/cal.txt 2>/tmp/cal_f.log
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/cal_w -- $R/tools/microbench/pmc_calibrate > /dev/null 2>/tmp/cal_w.log
cat /tmp/cal.txt
python - <<P
import csv, glob
def load(d, name):
    out = {}
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                out.setdefault(r['Kernel_Name'].split('(')[0].replace('void ', ''), []).append(float(r['Counter_Value']))
    return out
fe, wr = load('/tmp/cal_f', 'FETCH_SIZE'), load('/tmp/cal_w', 'WRITE_SIZE')
print('kernel, FETCH_SIZE (KB as reported), WRITE_SIZE (KB as reported)')
for k in fe:
    print('%-22s %14.0f %14.0f' % (k, sum(fe[k]) / len(fe[k]), sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)))
P
