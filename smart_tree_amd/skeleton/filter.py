"""Outlier removal (reference smart_tree/skeleton/filter.py:6-11)."""
from __future__ import annotations

import torch

from .graph import BOUND_LT, SEARCH_CELL_DIV, knn


def outlier_removal(points: torch.Tensor, radii: torch.Tensor, nb_points: int = 4, seg_off: torch.Tensor = None) -> torch.Tensor:
    """Keep a point iff its nb_points nearest neighbours (itself included, d = 0) all exist and lie
    closer than the point's own radius: `(dists < radii) & (idxs != -1)` summed == nb_points.
    The strict per-point bound runs inside the search, so "all nb_points slots filled" is the test."""
    if points.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.bool, device=points.device)
    bound = radii.reshape(-1)
    # r = -1: the search radius max(radii) is reduced on the device (one host round trip less); cell = r / 8
    # seg_off (additive): `points` holds several independent clouds; neighbours are searched inside a point's own cloud
    idxs, _, _ = knn(points, points, K=nb_points, r=-1.0, bound=bound, bound_mode=BOUND_LT, cell=-SEARCH_CELL_DIV,
                     src_seg_off=seg_off, dest_seg_off=seg_off)
    return idxs[:, nb_points - 1] != -1
