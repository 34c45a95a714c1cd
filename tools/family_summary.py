"""Per-cloud kernel time by kernel family from a rocprofv3 kernel_stats.csv: python tools/family_summary.py <csv> <clouds>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
clouds = float(sys.argv[2])
FAM = [("k_sparse_conv", "sparse convs"), ("k_heads", "heads"), ("k_bk_", "brick rulebooks"), ("k_rb_", "hash rulebooks / row moves"),
       ("k_vx_", "voxelize"), ("k_sk_select", "branch selection"), ("k_sk_claim", "chip-wide claim"), ("k_sk_sssp", "SSSP rounds"),
       ("k_sk_", "other skeleton (preds, ancestors, init)"), ("k_knn", "neighbour searches"), ("k_grid", "grids"), ("k_cc_", "components"),
       ("k_csr", "adjacency"), ("k_comp", "component layout"), ("k_scan", "scans"), ("k_sort", "sorts"), ("k_post", "post-process"),
       ("k_asm", "assemble"), ("k_bbox", "centre"), ("k_centre", "centre"), ("at::", "torch glue"), ("__amd_rocclr", "copies / fills")]
tot, cnt = defaultdict(float), defaultdict(float)
for r in rows:
    name = re.sub(r"^void ", "", r["Name"])
    fam = next((f for key, f in FAM if key in name), "other")
    tot[fam] += float(r["TotalDurationNs"])
    cnt[fam] += int(r["Calls"])
print("kernel time per cloud %.1f us, launches per cloud %.1f" % (sum(tot.values()) / clouds / 1e3, sum(cnt.values()) / clouds))
for f in sorted(tot, key=lambda f: -tot[f]):
    print("%9.1f us/cloud %7.2f launches/cloud  %s" % (tot[f] / clouds / 1e3, cnt[f] / clouds, f))
