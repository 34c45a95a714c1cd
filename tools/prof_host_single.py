"""GPU box: where the HOST time of one cloud through Pipeline.process_cloud goes (cProfile over 30 calls), and how long the
GPU idles between kernels.   python tools/prof_host_single.py"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
c = sample_tree_cloud(1_000_000, seed=0)
cl = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
for _ in range(5):
    pipe.process_cloud(cloud=cl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    pipe.process_cloud(cloud=cl)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / 20:.3f} ms per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    pipe.process_cloud(cloud=cl)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
