"""GPU box (one-off robustness probe): pathological clouds of 1M points through Pipeline.process_cloud -- every point identical, a line, a
flat sheet, every point three times, two trees 9 km apart, a tree scaled to millimetres, NaN / infinite medial vectors -- each under a
wall-clock bound: no hang, no crash, a sane outcome.    python tools/probe_adversarial.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
c = sample_tree_cloud(1_000_000, seed=0)
base = c["xyz"]
rng = np.random.RandomState(0)
cases = {
    "every point identical": np.tile(base[:1], (1_000_000, 1)),
    "a line": np.stack([np.zeros(1_000_000), np.linspace(0, 10, 1_000_000), np.zeros(1_000_000)], 1),
    "a flat sheet": np.concatenate([rng.rand(1_000_000, 2) * 8, np.zeros((1_000_000, 1))], 1),
    "every point three times": np.concatenate([base[:333_333]] * 3),
    "two trees 9 km apart": np.concatenate([base[:500_000], base[:500_000] + np.array([9000.0, 0, 0])]),
    "a tree in millimetres": base * 1e-3,
}
for name, xyz in cases.items():
    xyz = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32)).to(dev)
    t0 = time.time()
    try:
        sk = pipe.process_cloud(cloud=Cloud(xyz=xyz, rgb=torch.zeros_like(xyz)))
        torch.cuda.synchronize()
        out = f"{len(sk.skeletons)} trees, {sum(len(t.branches) for t in sk.skeletons)} branches"
    except Exception as e:  # noqa: BLE001
        out = f"{type(e).__name__}: {str(e)[:110]}"
    print(f"{name:28s}: {out}  ({time.time() - t0:.2f} s)", flush=True)
sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
for bad in (float("nan"), float("inf")):
    mv = c["medial_vector"].copy()
    mv[::1000] = bad
    t0 = time.time()
    s = sk.forward(Cloud(xyz=torch.from_numpy(base).to(dev), medial_vector=torch.from_numpy(mv).to(dev), class_l=torch.zeros((len(mv), 1), device=dev)))
    n = sum(len(t.branches) for t in s.skeletons)
    torch.cuda.synchronize()
    print(f"{'1000 ' + str(bad) + ' medial vectors':28s}: {len(s.skeletons)} trees, {n} branches  ({time.time() - t0:.2f} s)", flush=True)
