"""Dev script: phase timing inside k_skeleton_components for the 1M-point bench cloud."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch, numpy as np
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
from smart_tree_amd.skeleton import graph as G
from smart_tree_amd.skeleton.filter import outlier_removal
from smart_tree_amd.skeleton.skeletonize import run_components
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
c = sample_tree_cloud(1_000_000, seed=0)
cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
cloud = pipe.preprocessing(cloud)
lc = pipe.model_inference.forward(cloud)
bc = lc.filter_by_class([0])
print("branch points", len(bc))
medial, radius = G.medial_points(bc.xyz, bc.medial_vector)
mask = outlier_removal(medial, radius.unsqueeze(1), 8)
bc = bc.filter(mask); medial, radius = medial[mask], radius[mask]
print("after outlier", len(bc), "radius q", torch.quantile(radius, torch.tensor([0,.5,.9,1.0], device=dev)).tolist())
g = G.nn_graph(medial, radius.clamp(min=0.02), K=16)
comps = g.connected_cugraph_components(32)
print("edges", g.edges.shape[0], "comps", comps.n_components, comps.comp_size[:8].tolist())
for bt in (256, 512, 1024):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = run_components(comps, medial, radius, bc.xyz[:,1].contiguous(), block_threads=bt)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"select block {bt}: total {dt*1e3:.2f} ms; stats {res.stats}; branches {int(res.n_branches[0])}")
import ctypes
from smart_tree_amd import _lib
ticks = torch.zeros(16, dtype=torch.int64, device=dev)
L = _lib.lib(); L.st_debug_set_ticks.argtypes=[ctypes.c_void_p]; L.st_debug_set_ticks(ticks.data_ptr())
res = run_components(comps, medial, radius, bc.xyz[:,1].contiguous())
torch.cuda.synchronize(); L.st_debug_set_ticks(None)
t = ticks.cpu().numpy()
print('select phases (us, 100MHz ticks): head', t[0]/100, 'cursor', t[1]/100, 'trace', t[2]/100, 'path+record', t[3]/100, 'claim+finish', t[4]/100, '| iterations', t[8], 'small', t[9], 'path verts', t[10], 'candidates', t[11])
from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = run_components(comps, medial, radius, bc.xyz[:,1].contiguous(), stages=STAGE_SSSP)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sssp+preds only: {dt*1e3:.2f} ms {res.stats}")
