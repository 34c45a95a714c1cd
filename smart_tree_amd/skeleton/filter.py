"""Outlier removal (reference smart_tree/skeleton/filter.py:6-11)."""
from __future__ import annotations

import torch

from .graph import BOUND_LT, _search_cell, knn


def outlier_removal(points: torch.Tensor, radii: torch.Tensor, nb_points: int = 4) -> torch.Tensor:
    """Keep a point iff its nb_points nearest neighbours (itself included, d = 0) all exist and lie
    closer than the point's own radius: `(dists < radii) & (idxs != -1)` summed == nb_points.
    The strict per-point bound runs inside the search, so "all nb_points slots filled" is the test."""
    if points.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.bool, device=points.device)
    r_max = torch.max(radii).item()
    bound = radii.reshape(-1)
    idxs, _, _ = knn(points, points, K=nb_points, r=r_max, bound=bound, bound_mode=BOUND_LT, cell=_search_cell(bound, r_max))
    return idxs[:, nb_points - 1] != -1
