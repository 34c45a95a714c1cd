"""Developer probe: does the voxeliser's hash insert run faster when the points arrive in spatial order?  (Timing only: sorting
the points changes which point represents a voxel.)  python tools/probe_voxel_order.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import voxel_oracle as vo  # noqa: E402  (centre_cloud only: a developer tool, not product)
from smart_tree_amd.dataset.dataset import voxelize_blocks  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

dev = torch.device("cuda:0")
c = sample_tree_cloud(1_000_000, seed=0)
xyz = vo.centre_cloud(c["xyz"])
for name, order in (("input order (random)", None), ("sorted by 16 cm cell", 0.16), ("sorted by 4 cm cell", 0.04), ("sorted by 64 cm cell", 0.64)):
    p = xyz
    if order is not None:
        cell = np.floor((xyz - xyz.min(0)) / order).astype(np.int64)
        key = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
        p = xyz[np.argsort(key, kind="stable")]
    t = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
    rgb = torch.zeros_like(t)
    for _ in range(3):
        voxelize_blocks(t, rgb, 0.02)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        vb = voxelize_blocks(t, rgb, 0.02)
    b.record()
    torch.cuda.synchronize()
    print(f"{name:24s}: {a.elapsed_time(b) / 10 * 1e3:7.1f} us per call (whole voxelise stage incl. host round trips), {vb.coords.shape[0]} voxels")
