"""One skeleton branch: a poly-line with a radius per vertex.

Field names, order and semantics follow `smart_tree/data_types/branch.py:17-75`
(`length` :61-63, `initial_radius` :65-67).  xyz is [m,3] and radii [m,1] float32 on the host,
exactly what `sample_tree` (reference skeleton/path.py:128-133) constructs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from .tube import Tube


@dataclass
class BranchSkeleton:
    _id: int
    parent_id: int
    xyz: torch.Tensor
    radii: torch.Tensor
    child_id: Optional[int] = None

    def __post_init__(self):
        if self.xyz.ndim != 2 or self.xyz.shape[1] != 3:
            raise TypeError(f"xyz must be [N,3], got {tuple(self.xyz.shape)}")
        if self.radii.ndim != 2 or self.radii.shape != (self.xyz.shape[0], 1):
            raise TypeError(f"radii must be [{self.xyz.shape[0]},1], got {tuple(self.radii.shape)}")

    def __len__(self) -> int:
        return self.xyz.shape[0]

    def __str__(self) -> str:
        return f"Branch {self._id} (parent {self.parent_id}): {len(self)} vertices"

    def to_tubes(self) -> List[Tube]:
        return [Tube(self.xyz[i], self.xyz[i + 1], self.radii[i], self.radii[i + 1]) for i in range(len(self) - 1)]

    def filter(self, mask) -> "BranchSkeleton":
        return BranchSkeleton(self._id, self.parent_id, self.xyz[mask], self.radii[mask], self.child_id)

    @property
    def length(self) -> torch.Tensor:
        return (self.xyz[1:] - self.xyz[:-1]).norm(dim=1).sum()

    @property
    def initial_radius(self) -> torch.Tensor:
        return torch.max(self.radii[0], self.radii[-1])

    @property
    def biggest_radius_idx(self) -> torch.Tensor:
        return torch.argmax(self.radii)

    @property
    def biggest_radius(self) -> torch.Tensor:
        return torch.max(self.radii)
