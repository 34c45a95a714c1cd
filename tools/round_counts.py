"""How many SSSP launches and select/claim launch pairs the 1M-point synthetic trees need (read-back batches of one
launch): the numbers the batch sizes in csrc/skeleton.hip are chosen from.  python tools/round_counts.py [n_seeds]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from smart_tree_amd.skeleton import tuning  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.skeleton import skeletonize  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
seen = []
orig = skeletonize.run_components


def spy(*a, **k):
    res = orig(*a, **k)
    seen.append(res.stats)
    return res


skeletonize.run_components = spy
with tuning.override({7: 1, 3: 1}):  # one launch per read-back: the counts are exact
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
        c = sample_tree_cloud(1_000_000, seed=seed)
        pipe.process_cloud(cloud=Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
        print(seed, seen[-1], flush=True)
