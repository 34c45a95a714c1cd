"""Reference-named wrappers for single graphs (smart_tree/skeleton/shortest_path.py:12-55): `shortest_paths`
returns (verts, preds, distance) like the reference's cugraph.sssp call, computed by the same
persistent kernel the Skeletonizer uses (csrc/skeleton.hip) on a one-component layout."""
from __future__ import annotations

import torch

from ..data_types.graph import Graph
from .graph import connected_components
from .skeletonize import STAGE_SSSP, STAGE_TREE_DISTANCE, run_components


def _single_component(points, edges, edge_weights):
    comps = connected_components(Graph(points, edges, edge_weights), minimum_vertices=0)
    return comps


def shortest_paths(root, edges, edge_weights, renumber=True, points=None, surface_y=None):
    """SSSP from `root` over the undirected weighted graph; only the root's component gets finite
    distances (as with cugraph).  `surface_y` defaults to an indicator that makes `root` the lowest point."""
    root = int(root)
    if root < 0:
        raise ValueError(f"shortest_paths: root {root} out of range")
    # vertices = everything any argument mentions: an isolated root (or trailing isolated vertices) has an id above the
    # largest edge end point, and cugraph.sssp accepts such a source.  `renumber` is accepted for call-site parity: vertex ids
    # are used as they are (the reference's default renumbering is an internal detail of cugraph).
    n = max(int(edges.max().item()) + 1 if edges.numel() else 0, root + 1, 0 if points is None else int(points.shape[0]),
            0 if surface_y is None else int(surface_y.shape[0]))
    dev = edges.device
    for name, t in (("points", points), ("surface_y", surface_y)):
        if t is not None and int(t.shape[0]) != n:
            raise ValueError(f"shortest_paths: {name} has {int(t.shape[0])} rows, the graph has {n} vertices")
    pts = points if points is not None else torch.zeros((n, 3), device=dev)
    comps = _single_component(pts, edges, edge_weights)
    ys = torch.ones(n, device=dev)
    ys[root] = 0.0
    res = run_components(comps, pts, torch.zeros(n, device=dev), ys if surface_y is None else surface_y, stages=STAGE_SSSP)
    order = comps.vert_order.long()
    dist = torch.full((n,), float("inf"), device=dev)
    pred = torch.full((n,), -1, dtype=torch.int64, device=dev)
    c = int((comps.new_id[root] >= comps.comp_off[1:].to(comps.new_id.dtype)).sum().item()) if comps.n_components else 0
    a, b = int(comps.comp_off[c]), int(comps.comp_off[c + 1])
    ids = order[a:b]
    dist[ids] = res.dist[a:b]
    local_pred = res.pred[a:b].long()
    pred[ids] = torch.where(local_pred >= 0, ids[local_pred.clamp(min=0)], local_pred)
    return torch.arange(n, device=dev), pred, dist
