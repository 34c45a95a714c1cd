"""`sample_tree` with the reference's signature (smart_tree/skeleton/path.py:49-140) for ONE component:
greedy branch extraction from a predecessor tree.  Runs the `st_sample_tree` stage of csrc/skeleton.hip
(the same kernels `Skeletonizer.forward` uses for all components at once)."""
from __future__ import annotations

import ctypes
from typing import Dict

import torch

from .. import _lib
from ..data_types.branch import BranchSkeleton

STAGE_SAMPLE = 4


def sample_tree(medial_pts: torch.Tensor, medial_radii: torch.Tensor, preds: torch.Tensor, distances: torch.Tensor,
                all_points: torch.Tensor = None, root_idx: int = 0, visualize: bool = False, pbar=None,
                block_threads: int = 0) -> Dict[int, BranchSkeleton]:
    """medial_pts [n,3], medial_radii [n,1] or [n], preds [n] (-1 at the root), distances [n] (from the root).
    Returns {branch_id: BranchSkeleton} with host tensors, exactly as path.py:128-133 builds them."""
    L = _lib.lib()
    dev = medial_pts.device
    n = medial_pts.shape[0]
    if n == 0:
        return {}
    pts = medial_pts.contiguous().float()
    rad = medial_radii.reshape(-1).contiguous().float()
    pred = preds.to(torch.int32).contiguous()
    dist = distances.contiguous().float()
    i32 = lambda k: torch.empty((max(k, 1),), dtype=torch.int32, device=dev)
    comp_off = torch.tensor([0, n], dtype=torch.int32, device=dev)
    sizes = (ctypes.c_int32 * 1)(n)
    root_local = torch.zeros(1, dtype=torch.int32, device=dev)
    bparent, boff, blen, nbr, verts, bof = i32(n), i32(n), i32(n), i32(1), i32(n), i32(n)
    ws = _lib.workspace(L.st_skeleton_workspace_bytes(n, 1), dev)
    r_max = float(rad.max().item())
    _lib.check(L.st_skeleton_components(1, _lib.ptr(comp_off), sizes, n, _lib.ptr(pts), _lib.ptr(rad), _lib.ptr(rad), None, None,
                                        None, float(max(r_max / 4.0, 1e-4)), STAGE_SAMPLE, int(block_threads), _lib.ptr(dist),
                                        _lib.ptr(pred), _lib.ptr(root_local), None, _lib.ptr(bparent), _lib.ptr(boff),
                                        _lib.ptr(blen), _lib.ptr(nbr), _lib.ptr(verts), _lib.ptr(bof), None, _lib.ptr(ws),
                                        ws.numel(), _lib.stream(dev)))
    nb = int(nbr.item())
    parents, offs, lens, verts = bparent[:nb].tolist(), boff[:nb].tolist(), blen[:nb].tolist(), verts.long()
    pts_h, rad_h = pts.cpu(), rad.cpu()
    branches = {}
    for b in range(nb):
        ids = verts[offs[b]: offs[b] + lens[b]].cpu()
        branches[b] = BranchSkeleton(b, parents[b], xyz=pts_h[ids], radii=rad_h[ids].unsqueeze(1))
    return branches
