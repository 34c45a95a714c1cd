"""Developer aid: ms per process_cloud for the two bench clouds."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
from smart_tree_amd import profiling
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
for seed in (0, 1):
    c = sample_tree_cloud(1_000_000, seed=seed)
    cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
    for _ in range(3): pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize(); print('seed', seed, 'ms/step', (time.perf_counter() - t0) * 100)
