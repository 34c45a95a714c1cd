"""Cloud-level sharding across the GPUs of one node + the result gather.

The reference is single-GPU (`cuda:0` literals everywhere).  Tree clouds are independent units --
no shared state, BatchNorm in eval mode -- so a batch of clouds is split round-robin over the
ranks (one process per GPU) and every rank runs the whole pipeline on its own clouds with no
data-path collective.  The only communication is the variable-length gather of the finished
skeletons to rank 0: `all_gather` of the packed sizes (16 bytes per rank: every rank needs the padded
length), then ONE `gather` of the padded payloads to rank 0 only (KBs per cloud -- xGMI bandwidth is
irrelevant; RCCL when the backend is "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist

from .data_types.branch import BranchSkeleton
from .data_types.tree import DisjointTreeSkeleton, TreeSkeleton


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin ownership: item i belongs to rank i % world_size."""
    return list(range(rank, n_items, world_size))


def pack_skeleton(sk: DisjointTreeSkeleton, cloud_id: int = 0):
    """-> (table int64 [B,6] = (cloud, tree, branch, parent, offset, length), geom float32 [P,4] = (xyz, radius))."""
    fast = sk.pack(cloud_id) if hasattr(sk, "pack") else None  # DeviceSkeleton: straight from its packed host arrays
    if fast is not None:
        return fast
    rows, geom, off = [], [], 0
    for tree in sk.skeletons:
        for b in tree.branches.values():
            rows.append((cloud_id, tree._id, b._id, b.parent_id, off, len(b)))
            geom.append(torch.cat((b.xyz.float(), b.radii.reshape(-1, 1).float()), dim=1))
            off += len(b)
    table = torch.tensor(rows, dtype=torch.int64).reshape(-1, 6)
    return table, (torch.cat(geom) if geom else torch.zeros((0, 4), dtype=torch.float32))


def unpack_skeletons(table: torch.Tensor, geom: torch.Tensor) -> dict:
    """Inverse of pack_skeleton for any number of clouds: {cloud_id: DisjointTreeSkeleton}."""
    clouds: dict = {}
    for cloud, tree, branch, parent, off, length in table.tolist():
        trees = clouds.setdefault(cloud, {})
        t = trees.setdefault(tree, TreeSkeleton(tree, {}))
        g = geom[off: off + length]
        t.branches[branch] = BranchSkeleton(branch, parent, g[:, :3].contiguous(), g[:, 3:4].contiguous())
    return {c: DisjointTreeSkeleton([trees[k] for k in sorted(trees)]) for c, trees in clouds.items()}


def gather_skeletons(packed: Sequence, device=None, always_collective: bool = False):
    """Every rank passes [(table, geom), ...] for its clouds; rank 0 gets ([table], [geom]) per rank
    (others get None).  Works on any initialised process group; falls through when not distributed (or when the group
    has one rank, unless `always_collective`: tests/test_sharding.py runs the RCCL calls on a one-rank "nccl" group)."""
    tables = [t for t, _ in packed]
    geoms = [g for _, g in packed]
    # offsets in each table are relative to its own geom block: make them relative to the rank's block
    shift, fixed = 0, []
    for t, g in zip(tables, geoms):
        t = t.clone()
        t[:, 4] += shift
        shift += g.shape[0]
        fixed.append(t)
    table = torch.cat(fixed) if fixed else torch.zeros((0, 6), dtype=torch.int64)
    geom = torch.cat(geoms) if geoms else torch.zeros((0, 4), dtype=torch.float32)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always_collective):
        return [table], [geom]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    sizes = torch.tensor([table.shape[0], geom.shape[0]], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_rows = max(int(s[0]) for s in all_sizes)
    max_pts = max(int(s[1]) for s in all_sizes)
    pt = torch.zeros((max_rows, 6), dtype=torch.int64, device=dev)
    pg = torch.zeros((max_pts, 4), dtype=torch.float32, device=dev)
    pt[: table.shape[0]] = table.to(dev)
    pg[: geom.shape[0]] = geom.to(dev)
    # one payload per rank (the int64 table bit-cast behind the float32 geometry would save a call; two calls keep the
    # dtypes honest), received by rank 0 only
    out_t = [torch.zeros_like(pt) for _ in range(world)] if rank == 0 else None
    out_g = [torch.zeros_like(pg) for _ in range(world)] if rank == 0 else None
    dist.gather(pt, gather_list=out_t, dst=0)
    dist.gather(pg, gather_list=out_g, dst=0)
    if rank != 0:
        return None, None
    return ([out_t[r][: int(all_sizes[r][0])].cpu() for r in range(world)],
            [out_g[r][: int(all_sizes[r][1])].cpu() for r in range(world)])
