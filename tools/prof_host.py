"""Developer aid: cProfile of Pipeline.process_cloud on the bench cloud (host-side overheads)."""
import sys, cProfile, pstats
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
c = sample_tree_cloud(1_000_000, seed=0)
cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
for _ in range(3): pipe.process_cloud(cloud=cloud)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): pipe.process_cloud(cloud=cloud)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
