"""Developer aid: from a rocprofv3 kernel trace of the multi-stream bench, the fraction of wall time during which at
least one kernel runs, the average number of kernels in flight, and the same for the chip-filling kernels only."""
import csv, glob, sys
LAT = ('k_sk_select', 'k_sk_sssp', 'k_sk_claim', 'k_post')
fs = sorted(glob.glob('gpurun_out/prof_s8/*/*kernel_trace.csv'), key=lambda f: -len(open(f).readlines()))
rows = list(csv.DictReader(open(fs[0])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
t0, t1 = iv[0][0], max(x[1] for x in iv)
lo, hi = t0 + (t1 - t0) * 0.55, t0 + (t1 - t0) * 0.9   # steady state of the timed region
iv = [x for x in iv if lo <= x[0] and x[1] <= hi]
def cover(v):
    cov, cs, ce = 0, v[0][0], v[0][1]
    for s, e, *_ in v[1:]:
        if s > ce: cov += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return cov + ce - cs
wall = hi - lo
heads = sum(1 for x in iv if 'k_heads' in x[2])
print(f"{len(iv)} kernels on {len(set(x[3] for x in iv))} queues in {wall/1e6:.1f} ms, {heads} clouds -> {wall/1e6/max(heads,1):.2f} ms/cloud")
print(f"any kernel running: {cover(iv)/wall:.3f} of wall; kernels in flight on average: {sum(x[1]-x[0] for x in iv)/wall:.2f}")
big = [x for x in iv if not any(k in x[2] for k in LAT)]
print(f"non-single-workgroup kernels: cover {cover(big)/wall:.3f}, in flight {sum(x[1]-x[0] for x in big)/wall:.2f}")
gaps = []
ce = iv[0][1]
for s, e, *_ in iv[1:]:
    if s > ce: gaps.append(s - ce)
    ce = max(ce, e)
gaps.sort()
print(f"idle gaps: {len(gaps)}, total {sum(gaps)/1e6:.1f} ms, median {gaps[len(gaps)//2]/1e3:.1f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.1f} us" if gaps else "no gaps")
