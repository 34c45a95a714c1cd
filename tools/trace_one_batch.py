"""Developer aid: the kernels of ONE launch set, in launch order, from a rocprofv3 kernel trace of a single-stream bench run
(python tools/trace_one_batch.py <trace dir> [which batch from the end, default 1]).  Consecutive launches of the same kernel
are folded into one line (count, total, gaps).  The launch set is cut at k_bbox (first kernel of a batch: CentreCloud)."""
import csv, glob, re, sys
fs = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))
rows = list(csv.DictReader(open(fs[-1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
starts = [i for i, x in enumerate(iv) if x[2].startswith('k_bbox')]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = starts[-which]
hi = starts[-which + 1] if which > 1 else len(iv)
w = iv[lo:hi]
def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'at::native::', '', n)
    m = re.match(r'([A-Za-z_0-9:]+(<[^(]{0,40})?)', n)
    return (m.group(1) if m else n)[:70]
t0 = w[0][0]
print('launch set: %d kernels, wall %.3f ms, sum of durations %.3f ms' % (len(w), (w[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in w) / 1e6))
out, i = [], 0
while i < len(w):
    j = i
    while j + 1 < len(w) and short(w[j + 1][2]) == short(w[i][2]): j += 1
    dur = sum(e - s for s, e, _ in w[i:j + 1])
    gap = w[i][0] - w[i - 1][1] if i else 0
    print('%9.3f ms  +gap %7.1f us  x%-3d %9.1f us  %s' % ((w[i][0] - t0) / 1e6, gap / 1e3, j - i + 1, dur / 1e3, short(w[i][2])))
    i = j + 1
