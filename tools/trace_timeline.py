"""Developer aid: per-queue phase timeline of one pass (nb launch sets) of a bench.py kernel trace.
python tools/trace_timeline.py <trace dir> <launch sets per pass> <pass index>"""
import csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))
rows = list(csv.DictReader(open(fs[-1])))
qcol = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r[qcol]) for r in rows)
nb, which = int(sys.argv[2]), int(sys.argv[3])
starts = [i for i, x in enumerate(iv) if x[2].startswith('k_bbox')]
lo = starts[which * nb]
hi = starts[(which + 1) * nb] if (which + 1) * nb < len(starts) else len(iv)
w = iv[lo:hi]
PH = [('k_bbox', 'centre'), ('k_centre', 'centre'), ('k_vx_', 'voxelize'), ('k_bk_', 'bricks'), ('k_sparse_conv', 'convs'), ('k_rb_', 'convs'), ('k_heads', 'convs'),
      ('k_knn', 'knn'), ('k_grid', 'knn'), ('k_cc_', 'cc'), ('k_cl_', 'layout'), ('k_csr', 'csr'), ('k_sk_sssp', 'sssp'), ('k_sk_select', 'select'),
      ('k_sk_claim', 'select'), ('k_sk_', 'sk-other'), ('k_post', 'post'), ('k_asm', 'post')]
t0 = w[0][0]
for q in sorted(set(x[3] for x in w)):
    cur, a, b, busy, out = None, 0, 0, 0, []
    for s, e, n, qq in w:
        if qq != q: continue
        ph = next((p for k, p in PH if k in n), None)
        if ph is None: ph = cur or 'glue'
        if ph != cur:
            if cur: out.append((cur, a, b, busy))
            cur, a, busy = ph, s, 0
        b = e; busy += e - s
    if cur: out.append((cur, a, b, busy))
    print('queue', q)
    for ph, a, b, busy in out:
        if b - a > 50000: print('   %-9s %7.2f .. %7.2f ms  (%6.2f ms, busy %6.2f)' % (ph, (a - t0) / 1e6, (b - t0) / 1e6, (b - a) / 1e6, busy / 1e6))
