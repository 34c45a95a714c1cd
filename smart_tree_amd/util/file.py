"""Cloud / skeleton file I/O without open3d (reference smart_tree/util/file.py:73-167, .npz paths only)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from ..data_types.cloud import Cloud


def load_cloud(path) -> Cloud:
    """`.npz` clouds with the reference's keys (file.py:156-167 -> Cloud.from_numpy, cloud.py:233-252)."""
    path = Path(path)
    if path.suffix != ".npz":
        raise ValueError(f"only .npz clouds are supported without open3d, got {path}")
    with np.load(path) as z:
        cloud = Cloud.from_numpy(**{k: z[k] for k in z.files})
    cloud.filename = path
    return cloud


def save_cloud(path, cloud: Cloud) -> None:
    fields = {k: getattr(cloud, k) for k in ("xyz", "rgb", "medial_vector", "class_l", "branch_ids", "branch_direction")}
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in fields.items() if v is not None})


def save_skeleton_npz(path, skeleton) -> None:
    """Flat arrays: per branch (tree id, branch id, parent id, offset, length) + concatenated xyz / radii."""
    rows, xyz, radii, off = [], [], [], 0
    for tree in skeleton.skeletons:
        for b in tree.branches.values():
            rows.append((tree._id, b._id, b.parent_id, off, len(b)))
            xyz.append(b.xyz.numpy())
            radii.append(b.radii.reshape(-1).numpy())
            off += len(b)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    np.savez(path, branches=np.asarray(rows, dtype=np.int64).reshape(-1, 5),
             xyz=np.concatenate(xyz) if xyz else np.zeros((0, 3), np.float32),
             radii=np.concatenate(radii) if radii else np.zeros((0,), np.float32))
