"""Outlier removal (reference smart_tree/skeleton/filter.py:6-11)."""
from __future__ import annotations

import torch

from .graph import BOUND_LT, SEARCH_CELL_DIV, knn


def outlier_removal(points: torch.Tensor, radii: torch.Tensor, nb_points: int = 4, seg_off: torch.Tensor = None,
                    valid: torch.Tensor = None) -> torch.Tensor:
    """Keep a point iff its nb_points nearest neighbours (itself included, d = 0) all exist and lie
    closer than the point's own radius: `(dists < radii) & (idxs != -1)` summed == nb_points.
    The strict per-point bound runs inside the search, so "all nb_points slots filled" is the test.
    valid (additive, bool [n]): the filter runs over the points with valid[i] only -- exactly the result of filtering the array
    first and scattering the mask back (False elsewhere) -- so that a caller with a pending selection (the class filter) needs ONE
    compaction, and one host round trip, for both."""
    if points.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.bool, device=points.device)
    bound = radii.reshape(-1).contiguous().float()
    if valid is not None and nb_points != 8:  # (the general search has no subset form: filter, search, scatter back)
        idx = valid.nonzero().view(-1)
        sub_off = None if seg_off is None else torch.searchsorted(idx, seg_off.to(idx.dtype)).to(torch.int32)
        out = torch.zeros(points.shape[0], dtype=torch.bool, device=points.device)
        out[idx] = outlier_removal(points.index_select(0, idx), bound.index_select(0, idx), nb_points, sub_off)
        return out
    if nb_points != 8:  # the counting kernel is instantiated for the pipeline's nb_points; anything else takes the search
        idxs, _, _ = knn(points, points, K=nb_points, r=-1.0, bound=bound, bound_mode=BOUND_LT, cell=-SEARCH_CELL_DIV,
                         src_seg_off=seg_off, dest_seg_off=seg_off)
        return idxs[:, nb_points - 1] != -1
    # r = -1: the search radius max(radii) is reduced on the device (one host round trip less); cell = r / 8.
    # seg_off (additive): `points` holds several independent clouds; neighbours are counted inside a point's own cloud.
    # st_radius_count_seg = the same search, but it only counts and stops at the nb_points-th hit (no neighbour lists).
    from .. import _lib
    from . import tuning

    L = _lib.lib()
    pts = points.contiguous().float()
    n, dev = pts.shape[0], pts.device
    nseg = 1 if seg_off is None else int(seg_off.shape[0]) - 1
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    ws = _lib.workspace(L.st_knn_workspace_bytes_seg(n, nseg), dev)
    v8 = None if valid is None else valid.to(torch.uint8).contiguous()
    _lib.check(L.st_radius_count_seg(_lib.ptr(pts), n, _lib.ptr(pts), n, nb_points, -1.0, _lib.ptr(bound), BOUND_LT,
                                     -float(SEARCH_CELL_DIV), _lib.ptr(mask), _lib.ptr(seg_off), _lib.ptr(seg_off), nseg,
                                     _lib.ptr(ws), ws.numel(), _lib.stream(dev), tuning.count_cell_mean_mult(), _lib.ptr(v8)))
    return mask.bool()
