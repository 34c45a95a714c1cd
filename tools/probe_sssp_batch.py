"""GPU box: the SSSP stage (frontier launches + predecessors) of a BATCH of clouds alone, under tuning settings.
    python tools/probe_sssp_batch.py <clouds> ["k=v,.." ...]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.skeleton import graph as G  # noqa: E402
from smart_tree_amd.skeleton import tuning  # noqa: E402
from smart_tree_amd.skeleton.filter import outlier_removal  # noqa: E402
from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP, run_components  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

import ctypes  # noqa: E402
import os  # noqa: E402

from smart_tree_amd import _lib  # noqa: E402

if os.environ.get("ST_PROBE_LIB"):  # A/B aid: another build of the library (e.g. the previous revision's .so kept under _ab/)
    _lib._LIB = _lib.declare(ctypes.CDLL(os.environ["ST_PROBE_LIB"]))
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
clouds = []
for s in range(nb):
    c = sample_tree_cloud(1_000_000, seed=s)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
batch = pipe.preprocessing(Cloud.collate(clouds) if nb > 1 else clouds[0])
bc = pipe.model_inference.forward(batch).to_device(dev).filter_by_class([0])
medial, radius = G.medial_points(bc.xyz, bc.medial_vector)
keep = outlier_removal(medial, radius.unsqueeze(1), 8, seg_off=bc.seg_off).nonzero().view(-1)
bc = bc.filter(keep, assume_sorted=True)
medial, radius = medial.index_select(0, keep), radius.index_select(0, keep)
g = G.nn_graph(medial, radius.clamp(min=0.02), K=16, seg_off=bc.seg_off)
comps = g.connected_cugraph_components(32)
ys = bc.xyz[:, 1].contiguous()
print(f"{nb} clouds: {len(bc)} graph vertices, {comps.n_components} components")
for params in (sys.argv[2:] or [""]):
    knobs = {}
    for kv in filter(None, params.split(",")):
        k, v = kv.split("=")
        knobs[int(k)] = int(v)
    best = 1e9
    with tuning.override(knobs):
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = run_components(comps, medial, radius, ys, stages=STAGE_SSSP)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    full = 1e9
    with tuning.override(knobs):
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res2 = run_components(comps, medial, radius, ys)
            torch.cuda.synchronize()
            full = min(full, time.perf_counter() - t0)
    print(f"params [{params}]: SSSP + predecessors {best * 1e3:.3f} ms (best of 6), rounds {res.stats['sssp_rounds']}; with the branch selection "
          f"{full * 1e3:.3f} ms ({res2.stats['select_launches']} select launch(es), {res2.stats['branches']} branches)", flush=True)
