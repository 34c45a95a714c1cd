"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in float32 numpy, of the reference's blocking + voxelisation + collate stage:

  * CentreCloud                 smart_tree/dataset/augmentations.py:38-41, data_types/cloud.py:222-227
  * compute_blocks              smart_tree/dataset/dataset.py:166-190
  * cube_filter / torch_bb_filter  smart_tree/util/maths.py:135-155
  * SingleTreeInference.__getitem__  smart_tree/dataset/dataset.py:192-226
  * batch_collate               smart_tree/model/sparse.py:40-61   (block index -> coords[:,0])

Third-party arithmetic restated here (source NOT under /root/reference -- parity UNPINNED):
  spconv `PointToVoxel.generate_voxel_with_id` (spconv-cu117, unpinned; CPU path because the
  reference passes no device): voxel c = floorf((p - lo) / v) per axis in float32, grid =
  roundf((hi - lo) / v), points with c < 0 or c >= grid are dropped, with
  max_num_points_per_voxel = 1 the FIRST point in input order defines a voxel, voxels are
  numbered in order of first appearance, coordinates are emitted (z, y, x).
Canonical choices where the reference is random: blocks are emitted in `torch.unique(dim=0)`
order (lexicographic in x,y,z block id) instead of the DataLoader's accidental shuffle
(dataset.py:242 passes num_workers as `shuffle`), and ALL blocks go into one batch (BatchNorm
is in eval mode, blocks never interact, so batching does not change any value).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def centre_cloud(xyz: np.ndarray) -> np.ndarray:
    """augmentations.py:38-41: translate by -bbox_centre + [0, half_height, 0] (float32)."""
    xyz = xyz.astype(F32)
    lo, hi = xyz.min(0), xyz.max(0)
    half = (hi - lo) / F32(2)
    centre = lo + half
    shift = -centre + np.array([0, half[1], 0], dtype=F32)
    return xyz + shift


def compute_blocks(xyz: np.ndarray, block_size: float = 4.0, min_points: int = 20):
    """dataset.py:166-176.  Returns (block_ids [B,3] float32 sorted lexicographically, centres [B,3])."""
    q = np.floor(xyz.astype(F32) / F32(block_size))
    ids, counts = np.unique(q, axis=0, return_counts=True)
    ids = ids[counts > min_points]
    centres = ids * F32(block_size) + F32(block_size / 2)
    return ids.astype(F32), centres.astype(F32)


def cube_mask(points: np.ndarray, centre: np.ndarray, cube_size: float) -> np.ndarray:
    """maths.py:145-155 with :135-142: lo <= p < hi, lo/hi rounded to float32."""
    half = F32(cube_size / 2)
    lo = centre.astype(F32) - half
    hi = centre.astype(F32) + half
    return np.all((points >= lo) & (points < hi), axis=1)


def voxelize_block(pts6: np.ndarray, voxel_size: float):
    """dataset.py:196-222 + spconv PointToVoxel (CPU).  pts6 = [n,6] (xyz,rgb) float32.

    Returns (first_point_index [M] int64 ascending, coords_zyx [M,3] int32).
    """
    xyz = pts6[:, :3].astype(F32)
    v = F32(voxel_size)
    lo, hi = xyz.min(0), xyz.max(0)
    ext = (hi - lo) / v
    fl = np.floor(ext)
    grid = (fl + ((ext - fl) >= F32(0.5))).astype(np.int64)  # roundf (half away from zero), ext >= 0
    c = np.floor((xyz - lo) / v).astype(np.int64)
    ok = np.all((c >= 0) & (c < grid), axis=1)
    idx = np.nonzero(ok)[0]
    c = c[idx]
    # first point in input order wins; voxels numbered in order of first appearance
    key = (c[:, 0] * (grid[1] + 1) + c[:, 1]) * (grid[2] + 1) + c[:, 2]
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    return idx[first], c[first][:, ::-1].astype(np.int32)


def voxelize_cloud(xyz: np.ndarray, rgb: np.ndarray, voxel_size: float, block_size: float = 4.0,
                   buffer_size: float = 0.4, min_points: int = 20):
    """Whole inference-side data path: blocks -> per-block voxelisation -> one collated batch.

    Returns dict:
      feats  [M,6] float32  representative point (xyz,rgb) of each voxel
      coords [M,4] int32    (block, z, y, x)
      mask   [M]   bool     representative point inside the un-buffered block (dataset.py:224)
      point  [M]   int64    index into the input cloud of the representative point
      centres [B,3], block_lo [B,3] (voxel origin of each block = min over its halo points)
    """
    xyz = xyz.astype(F32)
    pts6 = np.concatenate([xyz, rgb.astype(F32)], axis=1)
    _, centres = compute_blocks(xyz, block_size, min_points)
    feats, coords, masks, point, los = [], [], [], [], []
    for b, centre in enumerate(centres):
        sel = np.nonzero(cube_mask(xyz, centre, block_size + buffer_size * 2))[0]
        block = pts6[sel]
        first, czyx = voxelize_block(block, voxel_size)
        f = block[first]
        feats.append(f)
        coords.append(np.concatenate([np.full((len(first), 1), b, np.int32), czyx], axis=1))
        masks.append(cube_mask(f[:, :3], centre, block_size))
        point.append(sel[first])
        los.append(block[:, :3].min(0) if len(block) else np.zeros(3, F32))
    cat = lambda xs, shape, dt: np.concatenate(xs) if xs else np.zeros(shape, dt)
    return {
        "feats": cat(feats, (0, 6), F32),
        "coords": cat(coords, (0, 4), np.int32),
        "mask": cat(masks, (0,), bool),
        "point": cat(point, (0,), np.int64),
        "centres": centres,
        "block_lo": np.array(los, dtype=F32).reshape(-1, 3),
    }
