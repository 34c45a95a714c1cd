"""Skeletonizer: branch cloud -> DisjointTreeSkeleton (reference smart_tree/skeleton/skeletonize.py:18-95).

Same constructor and `forward(cloud)` contract.  The reference walks the components in a host
loop (`process_subgraph`, :57-95: cugraph sub-graph -> pandas edge list -> SSSP -> predecessor
graph -> second SSSP -> `sample_tree`); here every component is processed by one launch of
`st_skeleton_components` (csrc/skeleton.hip), one persistent workgroup per component, and the
branches come back in a single device-to-host copy.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List

import torch

from .. import _lib, profiling
from ..data_types.branch import BranchSkeleton
from ..data_types.cloud import Cloud
from ..data_types.tree import DisjointTreeSkeleton, TreeSkeleton
from .filter import outlier_removal
from .graph import ComponentSet, medial_points, nn_graph

STAGE_SSSP, STAGE_TREE_DISTANCE, STAGE_SAMPLE = 1, 2, 4


@dataclass
class ComponentResult:
    """Device arrays in the renumbered vertex space of a ComponentSet (slices per component)."""
    dist: torch.Tensor
    pred: torch.Tensor
    root_local: torch.Tensor
    tree_dist: torch.Tensor
    branch_parent: torch.Tensor
    branch_off: torch.Tensor
    branch_len: torch.Tensor
    n_branches: torch.Tensor
    path_verts: torch.Tensor
    branch_of: torch.Tensor
    stats: dict = None


def run_components(comps: ComponentSet, medial_pts: torch.Tensor, radius: torch.Tensor, surface_y: torch.Tensor,
                   stages: int = STAGE_SSSP | STAGE_SAMPLE, block_threads: int = 0) -> ComponentResult:
    """SSSP from the lowest surface point, canonical predecessor tree, greedy branch extraction for
    every component.  The reference's second SSSP over the predecessor tree (skeletonize.py:80-85)
    re-adds the same float32 edge lengths in the same order and therefore reproduces the first
    SSSP's distances bit for bit (tests/test_skeleton.py proves it with STAGE_TREE_DISTANCE); the
    default pipeline feeds `dist` straight into sample_tree."""
    L = _lib.lib()
    dev = medial_pts.device
    C, m = comps.n_components, comps.vert_order.shape[0]
    order = comps.vert_order.long()
    pts = medial_pts[order].contiguous().float()
    rad = radius[order].contiguous().float()
    ys = surface_y[order].contiguous().float()
    f32 = lambda k: torch.empty((max(k, 1),), dtype=torch.float32, device=dev)
    i32 = lambda k: torch.empty((max(k, 1),), dtype=torch.int32, device=dev)
    res = ComponentResult(f32(m), i32(m), i32(C), f32(m), i32(m), i32(m), i32(m), i32(C), i32(m), i32(m))
    if C == 0:
        return res
    r_max = rad.max().item()
    sizes = comps.comp_size.cpu().numpy().astype("int32")  # host copy: sizes the claim grid
    stats = (ctypes.c_int64 * 4)()
    ws = _lib.workspace(L.st_skeleton_workspace_bytes(m, C), dev)
    n_adj = int(comps.row_off[-1].item()) if profiling.enabled() else 0
    with profiling.kernel("skeleton_stage", n_adj * 8 + m * (8 + 24)):
        _lib.check(L.st_skeleton_components(
            C, _lib.ptr(comps.comp_off.contiguous()), sizes.ctypes.data, m, _lib.ptr(pts), _lib.ptr(rad), _lib.ptr(ys),
            _lib.ptr(comps.row_off), _lib.ptr(comps.col), _lib.ptr(comps.wgt), float(max(r_max / 4.0, 1e-4)), int(stages),
            int(block_threads), _lib.ptr(res.dist), _lib.ptr(res.pred), _lib.ptr(res.root_local), _lib.ptr(res.tree_dist),
            _lib.ptr(res.branch_parent), _lib.ptr(res.branch_off), _lib.ptr(res.branch_len), _lib.ptr(res.n_branches),
            _lib.ptr(res.path_verts), _lib.ptr(res.branch_of), stats, _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    res.stats = {"sssp_rounds": stats[0], "plateau_rounds": stats[1], "branch_iterations": stats[2], "lift_levels": stats[3]}
    return res


class Skeletonizer:
    def __init__(self, K: int, min_connection_length: float, minimum_graph_vertices: int,
                 device: torch.device = torch.device("cuda:0")):
        self.K = K
        self.min_connection_length = min_connection_length
        self.minimum_graph_vertices = minimum_graph_vertices
        self.device = device
        self.block_threads = 0  # 0 = library default (1024 lanes in the per-component select workgroup)

    def forward(self, cloud: Cloud) -> DisjointTreeSkeleton:
        with profiling.stage("outlier_removal"):
            medial, radius = medial_points(cloud.xyz, cloud.medial_vector)
            mask = outlier_removal(medial, radius.unsqueeze(1), nb_points=8)
            cloud = cloud.filter(mask)
            medial, radius = medial[mask], radius[mask]
        with profiling.stage("nn_graph"):
            graph = nn_graph(medial, radius.clamp(min=self.min_connection_length), K=self.K)
        with profiling.stage("components"):
            comps = graph.connected_cugraph_components(minimum_vertices=self.minimum_graph_vertices)
        with profiling.stage("sssp_sample_tree"):
            res = run_components(comps, medial, radius, cloud.xyz[:, 1].contiguous(), block_threads=self.block_threads)
        with profiling.stage("assemble"):
            trees = self._assemble(comps, res, medial, radius)
        return DisjointTreeSkeleton(trees)

    def process_subgraph(self, cloud: Cloud, subgraph, skeleton_id: int = 0) -> TreeSkeleton:
        """Reference entry point (skeletonize.py:57-95) for ONE component of a ComponentSet."""
        comps, c = subgraph
        medial, radius = medial_points(cloud.xyz, cloud.medial_vector)
        res = run_components(comps, medial, radius, cloud.xyz[:, 1].contiguous(), block_threads=self.block_threads)
        return self._assemble(comps, res, medial, radius)[c]

    @staticmethod
    def _assemble(comps: ComponentSet, res: ComponentResult, medial: torch.Tensor, radius: torch.Tensor) -> List[TreeSkeleton]:
        """One D->H copy of the branch tables, then BranchSkeleton objects as path.py:128-133 builds them
        (xyz = medial_pts[path], radii = medial_radii[path] as [m,1], both on the host)."""
        C = comps.n_components
        if C == 0:
            return []
        nb = res.n_branches[:C].cpu().tolist()
        off = comps.comp_off.cpu().tolist()
        # flat branch table on the device: (component, branch) -> global path slice, then ONE gather + ONE copy
        rows = [(c, b) for c in range(C) for b in range(nb[c])]
        if not rows:
            return [TreeSkeleton(c, {}) for c in range(C)]
        dev = medial.device
        comp_of = torch.tensor([r[0] for r in rows], device=dev)
        slot = torch.tensor([off[c] + b for c, b in rows], device=dev)
        base = torch.tensor([off[c] for c, _ in rows], device=dev)
        lens = res.branch_len[slot].long()
        starts = base + res.branch_off[slot].long()
        total = int(lens.sum().item())
        seg = torch.repeat_interleave(torch.arange(len(rows), device=dev), lens, output_size=total)
        first = torch.cumsum(lens, 0) - lens
        pos = torch.arange(total, device=dev) - first[seg] + starts[seg]
        ids = comps.vert_order.long()[res.path_verts[pos].long() + base[seg]]
        geom = torch.cat((medial[ids], radius[ids].unsqueeze(1)), dim=1).cpu()
        parents = res.branch_parent[slot].cpu().tolist()
        pieces = torch.split(geom, lens.cpu().tolist())
        trees = [TreeSkeleton(c, {}) for c in range(C)]
        for (c, b), parent, g in zip(rows, parents, pieces):
            trees[c].branches[b] = BranchSkeleton(b, parent, xyz=g[:, :3].contiguous(), radii=g[:, 3:4].contiguous())
        return trees
