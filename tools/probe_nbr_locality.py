"""GPU box: how local are the submanifold neighbour tables in the network's row order?  For every level: the fraction of (offset, row)
pairs whose input row lies inside a window of W rows around the 128-row tile of its output row.
    python tools/probe_nbr_locality.py [clouds=1]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.model import sparse_ops as ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipe = bench.build_pipeline(dev)
clouds = []
for b in range(B):
    c = sample_tree_cloud(1_000_000, seed=b)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
cloud = pipe.preprocessing(Cloud.collate(clouds) if B > 1 else clouds[0])
vb = voxelize_blocks(cloud.xyz, cloud.rgb, 0.02, seg_off=cloud.seg_off)
pyr, order = ops.brick_pyramid(vb.coords, 3, vb.block_centres.shape[0], int(round(4.8 / 0.02)) + 2, vb.blk_seg, vb.n_seg)
for lvl in range(4):
    nbr = pyr.subm[lvl]  # [27, N]
    N = nbr.shape[1]
    rows = torch.arange(N, device=dev, dtype=torch.int64)
    tile0 = (rows // 128) * 128
    valid = nbr >= 0
    P = int(valid.sum())
    line = f"L{lvl} N={N} pairs={P} ({P / N:.1f} per row):"
    for W in (128, 256, 384, 512, 1024):
        m = (W - 128) // 2
        inside = valid & (nbr.long() >= (tile0 - m)[None, :]) & (nbr.long() < (tile0 + 128 + m)[None, :])
        line += f"  W={W}: {int(inside.sum()) / P:.3f}"
    print(line)
