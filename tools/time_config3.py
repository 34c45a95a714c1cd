"""GPU box: stage breakdown of BASELINE.json configs[3] (5M-point dense canopy, 1 cm voxels) through Pipeline.process_cloud,
one cloud at a time -- where the 290 ms go.  Also the whole-cloud (no halo) mode of SURVEY 8f.2 on configs[1] and configs[3].
    python tools/time_config3.py [n_points] [voxel] [foliage]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from smart_tree_amd import profiling  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
fol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
dev = torch.device("cuda:0")
c = sample_tree_cloud(n, seed=3 if fol > 0 else 0, foliage_fraction=fol)
cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
for blocking in ("blocks", "whole"):
    pipe = bench.build_pipeline(dev, voxel=voxel)
    pipe.model_inference.blocking = blocking
    pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        sk = pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    lc = pipe.last_labelled_cloud
    nvox = len(lc._base) if hasattr(lc, "_base") else len(lc)
    profiling.enable(True)
    sk = pipe.process_cloud(cloud=cloud)
    st = profiling.stage_ms(1)
    profiling.enable(False)
    nb = sum(len(t.branches) for t in sk.skeletons)
    print(f"{n} points, voxel {voxel}, foliage {fol}, blocking={blocking}: {ms:.2f} ms per cloud, {nvox} voxels through the network, "
          f"{len(sk.skeletons)} trees, {nb} branches")
    print("   stages (ms, with event brackets):", st)
