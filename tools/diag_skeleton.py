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
c = sample_tree_cloud(1_000_000, seed=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
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
import ctypes
from smart_tree_amd import _lib
from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP
L = _lib.lib(); L.st_debug_set_ticks.argtypes = [ctypes.c_void_p]
def timed(**kw):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = run_components(comps, medial, radius, bc.xyz[:,1].contiguous(), **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt * 1e3, res
L.st_debug_set_skeleton_param(-1, 0)
base_ms, _ = timed(stages=STAGE_SSSP)
def phases(tag):
    ticks = torch.zeros(16, dtype=torch.int64, device=dev)
    L.st_debug_set_ticks(ticks.data_ptr())
    run_components(comps, medial, radius, bc.xyz[:,1].contiguous())
    torch.cuda.synchronize(); L.st_debug_set_ticks(None)
    t = ticks.cpu().numpy()
    print(f'  {tag} phases us: head {t[0]/100:.0f} rank {t[7]/100:.0f} prune {t[1]/100:.0f} walk+rows {t[2]/100:.0f} claim {t[3]/100:.0f} validate+commit {t[4]/100:.0f} one-mode {t[5]/100:.0f} local {t[6]/100:.0f} ({t[15]}) | rounds {t[8]} slots {t[13]} commits {t[12]} one-mode iters {t[9]} wide {t[14]} cand {t[11]}')
DEF = {0: 1000, 1: 1 << 18, 2: 32, 3: 16, 4: 0, 5: 1 << 20}
def setp(kw={}):
    for k, v in {**DEF, **kw}.items(): L.st_debug_set_skeleton_param(int(k), int(v))
import smart_tree_amd.skeleton.skeletonize as SKM
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for div in (4.0, 6.0, 8.0):
    SKM.GRID_DIV = div
    setp({})
    ms, res = timed()
    print(f"grid div {div}: total {ms:.2f} ms  (select part {ms - base_ms:.2f}); stats {res.stats}; branches {int(res.n_branches[0])}")
    phases(f"div {div}")
SKM.GRID_DIV = 4.0
setp()
for bt in (256, 512):
    ms, res = timed(block_threads=bt)
    print(f"select block {bt}: total {ms:.2f} ms; branches {int(res.n_branches[0])}")
