"""Developer aid: st_points_to_nearest_tube on a 1M-point cloud x M tubes (the cloud-labelling utility)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from smart_tree_amd.util.queries import nearest_tube_device
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
n = 1_000_000
for m in (500, 4000, 16000):
    a = torch.from_numpy((rng.rand(m, 3) * 4).astype(np.float32)).to(dev)
    b = a + torch.from_numpy(rng.normal(0, 0.15, (m, 3)).astype(np.float32)).to(dev)
    r1 = torch.from_numpy((0.01 + 0.1 * rng.rand(m)).astype(np.float32)).to(dev)
    r2 = r1 * 0.8
    pts = torch.from_numpy((rng.rand(n, 3) * 4).astype(np.float32)).to(dev)
    nearest_tube_device(pts, a, b, r1, r2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): nearest_tube_device(pts, a, b, r1, r2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    pairs = n * m
    # 39 float32 operations per pair counted as flops (3+5 dot, div, 2 clamp, 6 proj, 4 radius, 3 diff, 5 dot, sqrt, sub, abs, ...)
    print(f"n={n} m={m}: {ms:.3f} ms, {pairs / ms / 1e6:.1f} G pairs/s, {39 * pairs / ms / 1e9:.1f} TFLOP/s (39 flop/pair; fp32 vector peak 157 TF with packed FMA)")
