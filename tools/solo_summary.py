"""Developer aid: per-cloud kernel time from a single-stream rocprofv3 --stats run (gpurun_out/prof_solo)."""
import csv, glob, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
f = sorted(glob.glob('gpurun_out/prof_solo/runc/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
N = int([r for r in rows if 'k_heads' in r['Name']][0]['Calls'])
skip = ('k_sk_select', 'k_sk_sssp', 'k_sk_claim', 'k_post')
tot = sum(float(r['TotalDurationNs']) / N / 1e3 for r in rows if not any(s in r['Name'] for s in skip))
groups = {}
for r in rows:
    for g in ('k_vx_', 'k_rb_', 'k_sparse_conv', 'k_csr', 'k_cc_', 'k_knn', 'k_scan', 'k_sk_', 'k_grid', '__amd', 'at::', 'rocprim'):
        if g in r['Name']:
            groups[g] = groups.get(g, 0) + float(r['TotalDurationNs']) / N / 1e3
            break
for r in rows:
    if pat and pat in r['Name']:
        print(f"{r['Name'][:64]:64s} {int(r['Calls'])/N:6.1f} {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/N/1e3:8.1f} us/cloud")
print({k: round(v) for k, v in groups.items()}, "chip-wide", round(tot), "passes", N)
