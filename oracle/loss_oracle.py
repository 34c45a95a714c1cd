"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in float64 numpy, of the reference's evaluation-side losses:

  * compute_loss              smart_tree/model/loss.py:7-51
  * L1Loss                    smart_tree/model/loss.py:54-56   (torch.nn.L1Loss: mean absolute difference)
  * cosine_similarity_loss    smart_tree/model/loss.py:59-61   (torch.nn.CosineSimilarity(dim=1, eps=1e-8): every vector is
                                                                divided by max(|v|, eps) before the dot product)
  * dice_loss                 smart_tree/model/loss.py:64-78
  * focal_loss                smart_tree/model/loss.py:81-97   (gamma = 2)

Pinned by tests/golden/loss_vectors.npz: outputs of the reference's own loss.py (imported from /root/reference by
tools/make_goldens.py under this container's torch) on seeded inputs, with and without mask / vector class.
The reference computes in float32 with torch's reduction order; the HIP path carries its sums in float64 -- both are held to
this float64 restatement (tests: 1e-5 relative for the HIP kernel, 1e-4 for the float32 goldens).
"""
from __future__ import annotations

import numpy as np


def l1_loss(outputs, targets) -> float:
    o, t = np.asarray(outputs, np.float64).reshape(-1), np.asarray(targets, np.float64).reshape(-1)
    return float(np.mean(np.abs(o - t))) if o.size else float("nan")


def cosine_similarity_loss(outputs, targets, eps: float = 1e-8) -> float:
    o, t = np.asarray(outputs, np.float64), np.asarray(targets, np.float64)
    if o.shape[0] == 0:
        return float("nan")
    no = np.maximum(np.sqrt((o * o).sum(1)), eps)[:, None]
    nt = np.maximum(np.sqrt((t * t).sum(1)), eps)[:, None]
    return float(np.mean(1.0 - ((o / no) * (t / nt)).sum(1)))


def _log_softmax(z):
    z = np.asarray(z, np.float64)
    z = z - z.max(1, keepdims=True)
    return z - np.log(np.exp(z).sum(1, keepdims=True))


def focal_loss(outputs, targets, gamma: float = 2.0) -> float:
    if len(outputs) == 0:
        return float("nan")
    logpt = _log_softmax(outputs)[np.arange(len(outputs)), np.asarray(targets).reshape(-1).astype(np.int64)]
    pt = np.exp(logpt)
    return float(np.mean(-1.0 * (1.0 - pt) ** gamma * logpt))


def dice_loss(outputs, targets, smooth: float = 1.0) -> float:
    p = np.exp(_log_softmax(outputs))
    onehot = np.zeros_like(p)
    onehot[np.arange(len(p)), np.asarray(targets).reshape(-1).astype(np.int64)] = 1.0
    inter = (p * onehot).sum()
    return float(1.0 - (2.0 * inter + smooth) / (p.sum() + onehot.sum() + smooth))


def compute_loss(preds, targets, mask=None, target_radius_log=True, vector_class=None, class_loss="focal"):
    """loss.py:7-51 with L1 / cosine / (focal | dice).  preds: dict of arrays; targets [n,5]."""
    radius = np.asarray(preds["radius"], np.float64).reshape(-1)
    direction = np.asarray(preds["direction"], np.float64)
    class_l = np.asarray(preds["class_l"], np.float64)
    targets = np.asarray(targets)
    t_class = targets[:, -1].astype(np.int64)  # .long()
    t_dir, t_rad = targets[:, 1:-1].astype(np.float64), targets[:, 0].astype(np.float64)
    if mask is not None:
        m = np.asarray(mask, bool)
        radius, direction, class_l, t_rad, t_dir, t_class = radius[m], direction[m], class_l[m], t_rad[m], t_dir[m], t_class[m]
    if vector_class is not None:
        v = t_class == vector_class
        radius, direction, t_rad, t_dir = radius[v], direction[v], t_rad[v], t_dir[v]
    if target_radius_log:
        t_rad = np.log(t_rad)
    cls = focal_loss if class_loss == "focal" else dice_loss
    return {"radius": l1_loss(radius, t_rad), "direction": cosine_similarity_loss(direction, t_dir), "class_l": cls(class_l, t_class)}


def process_cloud(xyz, inputs, targets, voxel_size):
    """TreeDataset.process_cloud (smart_tree/dataset/dataset.py:82-138) after the augmentation: whole-cloud PointToVoxel with the
    cloud's own bounding box as range (oracle/voxel_oracle.voxelize_block), features of the representative point, batch column 0,
    loss mask of ones.  Returns (input_feats, target_feats, coords [M,4] int32, loss_mask, representative point index)."""
    from . import voxel_oracle as vo

    xyz = np.asarray(xyz, np.float32)
    first, czyx = vo.voxelize_block(xyz, voxel_size)
    coords = np.concatenate([np.zeros((len(first), 1), np.int32), czyx], axis=1)
    return np.asarray(inputs)[first], np.asarray(targets)[first], coords, np.ones(len(first), bool), first
