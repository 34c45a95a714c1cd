"""Half-open axis-aligned box tests (reference `smart_tree/util/maths.py:135-155`)."""
from __future__ import annotations

import torch


def bb_filter(points: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """lo <= p < hi on all three axes."""
    return ((points >= lo) & (points < hi)).all(dim=1)


def cube_filter(points: torch.Tensor, center: torch.Tensor, cube_size: float) -> torch.Tensor:
    """Points inside the half-open cube [center - size/2, center + size/2).

    `cube_size / 2` is a Python float; subtracting it from a float32 tensor rounds it to float32
    first, which is what the reference's `center - (cube_size / 2)` does (maths.py:146-147).
    """
    half = cube_size / 2
    center = center.to(points.device)
    return bb_filter(points, center - half, center + half)
