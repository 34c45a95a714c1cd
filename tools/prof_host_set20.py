"""GPU box: where the HOST time of one launch set of 20 clouds goes (cProfile over 10 calls of Pipeline.process_clouds).
   python tools/prof_host_set20.py [clouds]"""
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

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
clouds = []
for s in range(nb):
    c = sample_tree_cloud(1_000_000, seed=s)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
for _ in range(4):
    pipe.process_clouds(clouds)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    pipe.process_clouds(clouds)
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / 10:.3f} ms per launch set of {nb}")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    pipe.process_clouds(clouds)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
