"""Skeleton containers and the host-side post-processing (prune / repair / smooth).

Reference: `smart_tree/data_types/tree.py:20-204`.
  * prune  (:94-121): walk branches in insertion order; drop when the parent was not kept, when
    `length < min_length`, or when `initial_radius < min_radius`; the root (smallest key) always stays.
  * repair (:73-92): prepend to each branch the nearest point on its parent's tube chain
    (`pts_to_nearest_tube`, reference util/queries.py:107-133), duplicating the first radius.
  * smooth (:123-134): box filter with zero padding, only when `len > kernel`; radii change
    shape from [n,1] to [n] (quirk kept -- SURVEY.md section 8c).
  * DisjointTreeSkeleton.prune touches only skeleton 0 (:164-168).
These loops run on a few hundred tiny host tensors; the reference runs them on the host too.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List

import torch
import torch.nn.functional as F

from .branch import BranchSkeleton
from .tube import Tube


def offset_to_nearest_tube(pt: torch.Tensor, chain_xyz: torch.Tensor, chain_radii: torch.Tensor) -> torch.Tensor:
    """Vector from `pt` to its projection on the tube (of the chain a_i -> a_{i+1} with radii r_i -> r_{i+1}) that
    minimises |distance - interpolated radius| (reference util/queries.py:89-133 for N = 1)."""
    a, b = chain_xyz[:-1], chain_xyz[1:]
    r1, r2 = chain_radii[:-1], chain_radii[1:]
    ab = b - a
    ap = pt.reshape(1, 3).float() - a
    t = ((ap * ab).sum(1) / (ab * ab).sum(1)).clip(0.0, 1.0)
    proj = a + t.unsqueeze(1) * ab
    r = (1 - t) * r1 + t * r2
    dist = (proj - pt.reshape(1, 3)).square().sum(1).sqrt()
    return proj[torch.argmin(torch.abs(dist - r))] - pt.reshape(3)


@dataclass
class TreeSkeleton:
    _id: int
    branches: Dict[int, BranchSkeleton]

    def __len__(self) -> int:
        return len(self.branches)

    def __str__(self) -> str:
        return f"Tree skeleton {self._id}: {len(self)} branches"

    def to_tubes(self) -> List[Tube]:
        return [t for b in self.branches.values() for t in b.to_tubes()]

    def repair(self) -> None:
        """Vectorised over the parent's tube chain (no per-tube objects): same arithmetic as
        `pts_to_nearest_tube` for one query point."""
        known = set(b._id for b in self.branches.values())
        for branch in self.branches.values():
            if branch.parent_id not in known:
                continue
            parent = self.branches[branch.parent_id]
            if len(parent) < 2:
                continue
            v = offset_to_nearest_tube(branch.xyz[0], parent.xyz, parent.radii.reshape(-1))
            branch.xyz = torch.cat(((branch.xyz[0] + v).reshape(1, 3), branch.xyz))  # tree.py:89-91
            branch.radii = torch.cat((branch.radii[[0]], branch.radii))

    def prune(self, min_radius: float, min_length: float, root_id=None) -> "TreeSkeleton":
        root_id = min(self.branches.keys()) if root_id is None else root_id
        keep = {root_id: self.branches[root_id]}
        dropped = {}
        for key, branch in self.branches.items():
            orphan = branch.parent_id not in keep and branch._id != root_id
            if orphan or branch.length < min_length or branch.initial_radius < min_radius:
                dropped[key] = branch
            else:
                keep[key] = branch
        self.branches = keep
        return TreeSkeleton(0, dropped)

    def smooth(self, kernel_size: int = 5) -> None:
        box = torch.ones(1, 1, kernel_size) / kernel_size
        for branch in self.branches.values():
            if branch.radii.shape[0] > kernel_size:
                branch.radii = F.conv1d(branch.radii.reshape(1, 1, -1), box, padding="same").reshape(-1)

    @property
    def length(self) -> torch.Tensor:
        return torch.sum(torch.tensor([b.length for b in self.branches.values()]))

    @property
    def max_branch_id(self) -> int:
        return max(self.branches.keys())


@dataclass
class DisjointTreeSkeleton:
    skeletons: List[TreeSkeleton]

    def prune(self, min_radius, min_length) -> None:
        if self.skeletons:
            self.skeletons[0].prune(min_radius=min_radius, min_length=min_length)

    def repair(self) -> None:
        for s in self.skeletons:
            s.repair()

    def smooth(self, kernel_size: int = 7) -> None:
        for s in self.skeletons:
            s.smooth(kernel_size=kernel_size)
