"""Point -> tapered-tube projection (`TreeSkeleton.repair`, cloud labelling).

Reference: `smart_tree/util/queries.py:89-133` (`pts_to_nearest_tube_gpu`): project each point on
every tube axis (t clipped to [0,1]), interpolate the radius, and pick the tube that minimises
|distance - radius|; `skeleton_to_points` (:139-166) labels a whole cloud with it in host chunks.
Points on the GPU go through `st_points_to_nearest_tube` (csrc/queries.hip: all points x all tubes in one
launch, no N x M intermediates); host tensors (the N = 1 call of the base-class `repair`) keep the plain torch
expression.
"""
from __future__ import annotations

from typing import List

import torch

from ..data_types.tube import Tube, collate_tubes


def nearest_tube_device(pts: torch.Tensor, a: torch.Tensor, b: torch.Tensor, r1: torch.Tensor, r2: torch.Tensor):
    """(vector [N,3], tube index [N] int64, radius [N]) for device tensors: pts [N,3]; a, b [M,3]; r1, r2 [M]."""
    from .. import _lib

    L = _lib.lib()
    dev = pts.device
    f = lambda t: t.to(dev).float().contiguous()
    pts, a, b, r1, r2 = f(pts), f(a), f(b), f(r1).reshape(-1), f(r2).reshape(-1)
    n, m = pts.shape[0], a.shape[0]
    vec = torch.empty((n, 3), dtype=torch.float32, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    rad = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.check(L.st_points_to_nearest_tube(_lib.ptr(pts), n, _lib.ptr(a), _lib.ptr(b), _lib.ptr(r1), _lib.ptr(r2), m,
                                           _lib.ptr(vec), _lib.ptr(idx), _lib.ptr(rad), _lib.stream(dev)))
    return vec, idx, rad


def pts_to_nearest_tube(pts: torch.Tensor, tubes: List[Tube], device=None):
    """Returns (vector point->projection [N,3], tube index [N], interpolated radius [N])."""
    if device is not None:
        pts = pts.to(device)
    if pts.is_cuda:
        ct = collate_tubes(tubes)
        return nearest_tube_device(pts, ct.a, ct.b, ct.r1, ct.r2)
    ct = collate_tubes(tubes).to(pts.device)
    pts = pts.float()
    ab = ct.b - ct.a  # [M,3]
    ap = pts[:, None, :] - ct.a[None, :, :]  # [N,M,3]
    t = (torch.einsum("nmd,md->nm", ap, ab) / torch.einsum("md,md->m", ab, ab)).clip(0.0, 1.0)
    proj = ct.a[None] + torch.einsum("nm,md->nmd", t, ab)
    r = (1 - t) * ct.r1 + t * ct.r2
    dist = (proj - pts[:, None, :]).square().sum(2).sqrt()
    idx = torch.argmin(torch.abs(dist - r), 1)
    rows = torch.arange(pts.shape[0], device=pts.device)
    return proj[rows, idx] - pts, idx, r[rows, idx]


# reference name
pts_to_nearest_tube_gpu = pts_to_nearest_tube


def skeleton_to_points(pcd, skeleton, chunk_size: int = 4096, device=None):
    """Reference `skeleton_to_points` (queries.py:139-166): (distance to, radius of, vector to) the nearest tube of
    `skeleton` for every point of `pcd`, as numpy arrays.  `chunk_size` is accepted for compatibility: the kernel takes
    the whole cloud in one launch."""
    import numpy as np

    dev = torch.device(device) if device is not None and device != "gpu" else torch.device("cuda:0")
    branches = [b for b in skeleton.branches.values()] if hasattr(skeleton, "branches") else \
        [b for t in skeleton.skeletons for b in t.branches.values()]
    a = torch.cat([b.xyz[:-1] for b in branches])
    bb = torch.cat([b.xyz[1:] for b in branches])
    r1 = torch.cat([b.radii.reshape(-1)[:-1] for b in branches])
    r2 = torch.cat([b.radii.reshape(-1)[1:] for b in branches])
    pts = pcd.xyz if torch.is_tensor(pcd.xyz) else torch.as_tensor(np.asarray(pcd.xyz))
    vec, _, rad = nearest_tube_device(pts.to(dev), a, bb, r1, r2)
    vec = vec.cpu().numpy()
    return np.sqrt(np.einsum("ij,ij->i", vec, vec)), rad.cpu().numpy(), vec
