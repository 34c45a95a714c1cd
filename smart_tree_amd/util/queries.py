"""Point -> tapered-tube projection used by `TreeSkeleton.repair`.

Reference: `smart_tree/util/queries.py:89-133` (`pts_to_nearest_tube_gpu`): project each point on
every tube axis (t clipped to [0,1]), interpolate the radius, and pick the tube that minimises
|distance - radius|.  N x M dense; the reference calls it with N = 1 per branch, so this stays a
plain torch expression on whatever device the inputs live on.
"""
from __future__ import annotations

from typing import List

import torch

from ..data_types.tube import Tube, collate_tubes


def pts_to_nearest_tube(pts: torch.Tensor, tubes: List[Tube]):
    """Returns (vector point->projection [N,3], tube index [N], interpolated radius [N])."""
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
