"""GPU box: ONE cloud at a time through Pipeline.process_cloud (the unit SURVEY 8d's metric times): wall time per cloud in a
loop, stage brackets (HIP events), host-side time per stage (perf_counter around the same stages, no events).
    python tools/time_single.py [n_points] [voxel] [seed] [reps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from smart_tree_amd import profiling  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, voxel=voxel)
clouds = []
for s in range(seed, seed + 2):
    c = sample_tree_cloud(n, seed=s)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
for _ in range(3):
    for cl in clouds:
        pipe.process_cloud(cloud=cl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    for cl in clouds:
        sk = pipe.process_cloud(cloud=cl)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / (reps * len(clouds))
print(f"{n} points, voxel {voxel}: {ms:.3f} ms per cloud (one at a time, {reps * len(clouds)} calls); "
      f"{len(sk.skeletons)} trees, {sum(len(t.branches) for t in sk.skeletons)} branches")
profiling.enable(True)  # one discarded pass with the timers on: their first use pays one-time costs (device query, event pools)
for cl in clouds:
    pipe.process_cloud(cloud=cl)
profiling.stage_ms(len(clouds))
profiling.enable(False)
profiling.enable(True)
for cl in clouds:
    pipe.process_cloud(cloud=cl)
st = profiling.stage_ms(len(clouds))
profiling.enable(False)
print("   stage brackets (ms per cloud, HIP events):", st, " sum of top-level:",
      round(sum(v for k, v in st.items() if k in ("preprocess", "voxelize", "unet", "class_filter", "outlier_removal", "nn_graph",
                                                   "components", "sssp_sample_tree", "assemble", "post_process")), 3))
