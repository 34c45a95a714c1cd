"""GPU box: phase timing inside k_sk_select (tuning code 15: a device array of phase timers) for one cloud: python tools/diag_phases.py [n] [voxel] [foliage] [seed] [params "k=v,..."]"""
import ctypes
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402

import ctypes as _ct, os as _os
from smart_tree_amd import _lib as _l
if _os.environ.get("ST_PROBE_LIB"):  # A/B aid: another build of the library (e.g. the previous revision's .so kept under _ab/)
    _l._LIB = _l.declare(_ct.CDLL(_os.environ["ST_PROBE_LIB"]))
from smart_tree_amd import _lib  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.skeleton import graph as G  # noqa: E402
from smart_tree_amd.skeleton.filter import outlier_removal  # noqa: E402
from smart_tree_amd.skeleton import tuning  # noqa: E402
from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP, run_components  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
fol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
seed = int(sys.argv[4]) if len(sys.argv) > 4 else (3 if fol > 0 else 0)
params = sys.argv[5] if len(sys.argv) > 5 else ""
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, voxel=voxel)
c = sample_tree_cloud(n, seed=seed, foliage_fraction=fol)
cloud = pipe.preprocessing(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
bc = pipe.model_inference.forward(cloud).filter_by_class([0])
medial, radius = G.medial_points(bc.xyz, bc.medial_vector)
mask = outlier_removal(medial, radius.unsqueeze(1), 8)
bc = bc.filter(mask)
medial, radius = medial[mask], radius[mask]
g = G.nn_graph(medial, radius.clamp(min=0.02), K=16)
comps = g.connected_cugraph_components(32)
print(f"{n} points, voxel {voxel}, foliage {fol}: graph vertices {len(bc)}, components {comps.n_components}, largest {comps.comp_size[:4].tolist()}, "
      f"radius quantiles {torch.quantile(radius[:1000000], torch.tensor([0, .5, .9, 1.0], device=dev)).tolist()}")
L = _lib.lib()
knobs = {}
for kv in filter(None, params.split(",")):
    k, v = kv.split("=")
    knobs[int(k)] = int(v)
ys = bc.xyz[:, 1].contiguous()


def timed(**kw):
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with tuning.override(knobs):
            res = run_components(comps, medial, radius, ys, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt * 1e3, res


sssp_ms, _ = timed(stages=STAGE_SSSP)
ms, res = timed()
print(f"params [{params}]: SSSP + predecessors {sssp_ms:.2f} ms, with the branch selection {ms:.2f} ms; {res.stats}; branches of the first component {int(res.n_branches[0])}")
ticks = torch.zeros(32, dtype=torch.int64, device=dev)
with tuning.override({**knobs, tuning.TICKS: ticks.data_ptr()}):
    run_components(comps, medial, radius, ys)
torch.cuda.synchronize()
t = ticks.cpu().numpy()
print(f"  phases (us, summed over components): head {t[0]/100:.0f} rank {t[7]/100:.0f} prune {t[1]/100:.0f} walk+rows {t[2]/100:.0f} claim {t[3]/100:.0f} "
      f"validate+commit {t[4]/100:.0f} one-mode {t[5]/100:.0f} local {t[6]/100:.0f} ({t[15]}) | rounds {t[8]} slots {t[13]} commits {t[12]} "
      f"one-mode iters {t[9]} (path vertices {t[10]}) wide {t[14]} cand {t[11]}")
