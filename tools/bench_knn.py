"""Dev script: the two neighbour searches of the skeleton stage (outlier_removal's counting query, nn_graph's K = 16 search)
on the labelled branch points of a batch of bench clouds, timed alone.

    python tools/bench_knn.py [clouds=24]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.skeleton import skeletonize as sk
from smart_tree_amd.skeleton.filter import outlier_removal
from smart_tree_amd.skeleton.graph import medial_points, nn_graph
from smart_tree_amd.synthetic import sample_tree_cloud

import os
from smart_tree_amd.skeleton import graph as G
from smart_tree_amd.skeleton import filter as F
if os.environ.get("ST_CELL_DIV"): G.SEARCH_CELL_DIV = G.GRAPH_CELL_DIV = F.SEARCH_CELL_DIV = float(os.environ["ST_CELL_DIV"])
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
CACHE = Path(f"/tmp/bench_knn_{B}.pt")  # the captured input: experimental builds of the search need not survive the pipeline
clouds = []
for b in range(0 if CACHE.exists() else B):
    c = sample_tree_cloud(1_000_000, seed=b)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
if CACHE.exists():
    d = torch.load(CACHE)
    cloud = Cloud(xyz=d["xyz"].to(dev), medial_vector=d["mv"].to(dev))
    cloud.seg_off = d["seg_off"].to(dev)
else:
    seen = {}
    orig = sk.Skeletonizer.forward
    def capture(self, cloud):
        seen["cloud"] = cloud
        return orig(self, cloud)
    sk.Skeletonizer.forward = capture
    bench.build_pipeline(dev).process_clouds(clouds)
    cloud = seen["cloud"]
    torch.save({"xyz": cloud.xyz.cpu(), "mv": cloud.medial_vector.cpu(), "seg_off": cloud.seg_off.cpu()}, CACHE)
medial, radius = medial_points(cloud.xyz, cloud.medial_vector)

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

mask = outlier_removal(medial, radius.unsqueeze(1), nb_points=8, seg_off=cloud.seg_off)
t_count = timeit(lambda: outlier_removal(medial, radius.unsqueeze(1), nb_points=8, seg_off=cloud.seg_off))
keep = mask.nonzero().view(-1)
cl = cloud.filter(keep)
m2, r2 = medial.index_select(0, keep), radius.index_select(0, keep).clamp(min=0.02)
g = nn_graph(m2, r2, K=16, seg_off=cl.seg_off)
t_nn = timeit(lambda: nn_graph(m2, r2, K=16, seg_off=cl.seg_off))
valid = g.idxs >= 0
print(f"{B} clouds: {medial.shape[0]} labelled points, {m2.shape[0]} kept; counting query {t_count:.3f} ms, K=16 search {t_nn:.3f} ms "
      f"(grid build included); neighbours per row {valid.sum().item() / m2.shape[0]:.2f}, check {int((g.idxs * valid).sum())}")
