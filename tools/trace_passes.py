"""Developer aid: a rocprofv3 kernel trace of `bench.py --steps K` cut into passes of `nb` launch sets (a launch set starts with
k_bbox): wall time of each pass, how much of it the chip-filling kernels cover, and the exposed tail (last chip-filling kernel ->
last kernel).  python tools/trace_passes.py <trace dir> <launch sets per pass>"""
import csv, glob, sys
THIN = ('k_sk_', 'k_post_process', 'k_asm_', 'k_scan', 'k_sort', 'rocclr', 'at::', 'k_cl_', 'k_fill', 'k_grid_init', 'k_grid_dims', 'k_grid_ncell')
fs = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))
rows = list(csv.DictReader(open(fs[-1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
nb = int(sys.argv[2])
starts = [i for i, x in enumerate(iv) if x[2].startswith('k_bbox')]
def cover(v):
    if not v: return 0
    cov, cs, ce = 0, v[0][0], v[0][1]
    for s, e, *_ in v[1:]:
        if s > ce: cov += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return cov + ce - cs
for p in range(0, len(starts) - nb + 1, nb):
    lo = starts[p]
    hi = starts[p + nb] if p + nb < len(starts) else len(iv)
    w = iv[lo:hi]
    # a pass ends with its last skeleton / post-process kernel: drop what follows a long silence (the next pass's host work)
    end = max(e for s, e, n in w if 'k_post_process' in n or 'k_sk_' in n) if any('k_post_process' in n for _, _, n in w) else w[-1][1]
    w = [x for x in w if x[0] <= end]
    big = [x for x in w if not any(t in x[2] for t in THIN)]
    t0 = w[0][0]
    last_big = max(e for _, e, _ in big)
    print('pass %2d: wall %7.2f ms; chip-filling kernels cover %6.2f ms (sum %6.2f), any kernel %6.2f ms; exposed tail %6.2f ms; launches %d'
          % (p // nb, (end - t0) / 1e6, cover(big) / 1e6, sum(e - s for s, e, _ in big) / 1e6, cover(w) / 1e6, (end - last_big) / 1e6, len(w)))
