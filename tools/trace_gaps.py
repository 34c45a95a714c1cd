"""Developer aid: idle time of the chip-filling kernels in a rocprofv3 kernel trace of bench.py (gpurun_out/prof_trace).
Steady state = the stretch of the trace in which consecutive network launches (k_heads, one per batch) are evenly spaced."""
import csv, glob, sys
THIN = ('k_sk_', 'k_post_process', 'k_asm_')
fs = sorted(glob.glob((sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_trace') + '/*/*kernel_trace.csv'))
rows = list(csv.DictReader(open(fs[-1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
heads = [x[0] for x in iv if 'k_heads' in x[2]]
print(len(iv), 'kernels,', len(heads), 'batches; spacing of the network launches (ms):', [round((b - a) / 1e6, 1) for a, b in zip(heads, heads[1:])])
def cover(v):
    cov, cs, ce = 0, v[0][0], v[0][1]
    for s, e, *_ in v[1:]:
        if s > ce: cov += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return cov + ce - cs
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) // 2
for a, b in ((k, k + 4), (k + 4, k + 8)):
    if b >= len(heads): break
    lo, hi = heads[a], heads[b]
    w = [x for x in iv if lo <= x[0] and x[1] <= hi]
    big = [x for x in w if not any(t in x[2] for t in THIN)]
    thin = [x for x in w if any(t in x[2] for t in THIN)]
    wall = hi - lo
    print('batches %d..%d: %.1f ms = %.2f ms per batch; any kernel %.3f of wall; chip-filling kernels cover %.3f (sum of durations %.3f); skeleton kernels cover %.3f'
          % (a, b, wall / 1e6, wall / 1e6 / (b - a), cover(w) / wall, cover(big) / wall, sum(x[1] - x[0] for x in big) / wall, cover(thin) / wall))
    gaps, ce = [], big[0][1]
    for s, e, *_ in big[1:]:
        if s > ce: gaps.append(s - ce)
        ce = max(ce, e)
    gaps.sort(reverse=True)
    print('   gaps between chip-filling kernels: %d, total %.1f ms; > 50 us: %d totalling %.1f ms; largest (us): %s'
          % (len(gaps), sum(gaps) / 1e6, sum(1 for g in gaps if g > 5e4), sum(g for g in gaps if g > 5e4) / 1e6, [round(g / 1e3) for g in gaps[:16]]))
