"""Tapered tube segment types (reference `smart_tree/data_types/tube.py:9-50`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch


@dataclass
class Tube:
    a: torch.Tensor  # [3] start point
    b: torch.Tensor  # [3] end point
    r1: torch.Tensor  # [1] start radius
    r2: torch.Tensor  # [1] end radius


@dataclass
class CollatedTube:
    a: torch.Tensor  # [M,3]
    b: torch.Tensor  # [M,3]
    r1: torch.Tensor  # [1,M]
    r2: torch.Tensor  # [1,M]

    def to(self, device) -> "CollatedTube":
        return CollatedTube(self.a.to(device), self.b.to(device), self.r1.to(device), self.r2.to(device))


def collate_tubes(tubes: List[Tube]) -> CollatedTube:
    """Stack a list of tubes; r1/r2 become row vectors [1,M] (reference tube.py:43-50)."""
    stack = lambda name: torch.cat([getattr(t, name).reshape(-1) for t in tubes])
    return CollatedTube(stack("a").reshape(-1, 3), stack("b").reshape(-1, 3),
                        stack("r1").reshape(1, -1), stack("r2").reshape(1, -1))
