"""Neighbourhood graph of the medial points (reference smart_tree/skeleton/graph.py).

`knn` / `nn` keep the reference signatures and return conventions (graph.py:12-33): idx [n,K]
int64 with -1 padding, distances = sqrt of the squared FRNN distances (NaN where padded).  The
search itself is `st_knn_radius` (csrc/knn.hip); edge list, connected components and the
per-component renumbering are csrc/graph.hip.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib
from . import tuning
from ..data_types.graph import Graph, KnnGraph, PaddedGraph

BOUND_NONE, BOUND_LE, BOUND_LT = 0, 1, 2


def knn(src: torch.Tensor, dest: torch.Tensor, K: int = 50, r: float = 1.0, grid=None, bound: Optional[torch.Tensor] = None,
        bound_mode: int = BOUND_NONE, cell: float = 0.0, src_seg_off: Optional[torch.Tensor] = None,
        dest_seg_off: Optional[torch.Tensor] = None, keep_kernel_width: bool = False):
    """The <= K nearest `dest` points of every `src` point with d^2 < r^2, ascending (ties: lower index).

    `bound` (per query) additionally keeps only d <= bound[i] (BOUND_LE) or d < bound[i] (BOUND_LT);
    used by nn_graph / outlier_removal, whose own filters make that exact (csrc/knn.hip header).
    src_seg_off / dest_seg_off ([B+1] int32, device): src and dest hold B independent clouds each; a query only sees
    its own cloud (additive keywords, Cloud.collate).
    """
    if K not in (1, 8, 16, 32, 64):
        # the kernels hold 1, 8, 16, 32 or 64 neighbours per query; any other K <= 64 -- the reference's defaults K = 50 (graph.py:12)
        # and nn_graph's K = 40 (:36) among them -- is the first K columns of the next width (the rows are sorted by (distance,
        # index), so the K nearest of 64 are the K nearest).  keep_kernel_width: the table keeps that width with the columns
        # past K emptied (-1 / NaN) -- what the graph builders, which want a power of two, read
        if not 1 <= K <= 64:
            raise ValueError(f"knn: 1 <= K <= 64 (got {K}; the reference's skeletoniser uses 16)")
        idx, dist, grid = knn(src, dest, 8 if K < 8 else (16 if K < 16 else (32 if K < 32 else 64)), r, grid, bound, bound_mode, cell,
                              src_seg_off, dest_seg_off)
        if keep_kernel_width:
            idx[:, K:] = -1
            dist[:, K:] = float("nan")
            return idx, dist, grid
        return idx[:, :K].contiguous(), dist[:, :K].contiguous(), grid
    L = _lib.lib()
    dev = src.device
    src = src.contiguous().float()
    dest = dest.contiguous().float()
    n1, n2 = src.shape[0], dest.shape[0]
    idx = torch.empty((n1, K), dtype=torch.int64, device=dev)
    dist = torch.empty((n1, K), dtype=torch.float32, device=dev)
    if n1 == 0:
        return idx, dist, grid
    if n2 == 0:
        return idx.fill_(-1), dist.fill_(float("nan")), grid
    nseg = 1 if src_seg_off is None else int(src_seg_off.shape[0]) - 1
    if (dest_seg_off is None) != (src_seg_off is None):
        raise ValueError("knn: batched searches need the cloud offsets of both src and dest")
    ws = _lib.workspace(L.st_knn_workspace_bytes_seg(n2, nseg), dev)
    b = bound.contiguous().float() if bound is not None else None
    _lib.check(L.st_knn_radius_seg(_lib.ptr(src), n1, _lib.ptr(dest), n2, K, float(r), _lib.ptr(b), bound_mode, float(cell),
                                   _lib.ptr(idx), _lib.ptr(dist), _lib.ptr(src_seg_off), _lib.ptr(dest_seg_off), nseg,
                                   _lib.ptr(ws), ws.numel(), _lib.stream(dev), tuning.knn_cell_mean_mult()))
    return idx, dist, grid


def medial_points(xyz: torch.Tensor, medial_vector: torch.Tensor):
    """(xyz + medial_vector, |medial_vector|) with the float32 operation order fixed in csrc/graph.hip
    (`Cloud.medial_pts` / `Cloud.radius`, reference data_types/cloud.py:229-231,254-256)."""
    L = _lib.lib()
    xyz = xyz.contiguous().float()
    mv = medial_vector.contiguous().float()
    n = xyz.shape[0]
    medial = torch.empty_like(xyz)
    radius = torch.empty((n,), dtype=torch.float32, device=xyz.device)
    _lib.check(L.st_medial_points(_lib.ptr(xyz), _lib.ptr(mv), n, _lib.ptr(medial), _lib.ptr(radius), _lib.stream(xyz.device)))
    return medial, radius


def nn(src, dest, r=1.0, grid=None):
    idx, dist, grid = knn(src, dest, K=1, r=r, grid=grid)
    return idx.squeeze(1), dist.squeeze(1), grid


SEARCH_CELL_DIV = 8.0  # knn(..., r=-1, cell=-SEARCH_CELL_DIV): the same cell, with r_max left on the device
GRAPH_CELL_DIV = 12.0  # nn_graph's K-nearest search reads every candidate of its cells: finer cells pay (24 clouds: 2.21 -> 2.07 ms;
                       # 16: 2.28); outlier_removal's counting query stops at the 8th hit and prefers the coarser grid (1.28 / 1.31 ms)


def _search_cell(radii: torch.Tensor, r_max: float) -> float:
    """Grid cell for per-query bounded searches: fine enough for thin twigs, coarse enough that
    the thickest branch scans a bounded number of cells."""
    return max(r_max / 8.0, 1e-4)


def make_edges(dists: torch.Tensor, idxs: torch.Tensor, padded: bool = False, seg_off: Optional[torch.Tensor] = None):
    """graph.py:52-60: edges (i -> idx) for idx > 0 (the reference's filter drops every edge INTO
    vertex 0 and vertex 0's self loop; kept), in (i, k) order.  `padded`: the capacity-sized arrays come back as they
    are, tail filled with (0, 0) / 0, and the edge count is not read back (data_types.graph.PaddedGraph).  seg_off (batched
    clouds): "vertex 0" is the first vertex of the query's own cloud."""
    L = _lib.lib()
    dev = dists.device
    n, K = dists.shape
    edges = torch.empty((max(n * K, 1), 2), dtype=torch.int64, device=dev)
    w = torch.empty((max(n * K, 1),), dtype=torch.float32, device=dev)
    ne = ctypes.c_int64(0)
    ws = _lib.workspace(L.st_make_edges_workspace_bytes(n), dev)
    nseg = 1 if seg_off is None else int(seg_off.shape[0]) - 1
    _lib.check((L.st_make_edges_seg_nowait if padded else L.st_make_edges_seg)(
        _lib.ptr(idxs.contiguous()), _lib.ptr(dists.contiguous()), n, K, _lib.ptr(edges), _lib.ptr(w),
        None if padded else ctypes.byref(ne), _lib.ptr(seg_off), nseg, _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    if padded:
        return edges[: n * K], w[: n * K]
    return edges[: ne.value], w[: ne.value]


def nn_graph(points: torch.Tensor, radii: torch.Tensor, K: int = 40, seg_off: Optional[torch.Tensor] = None) -> Graph:
    """graph.py:36-40: kNN with r = max radius, neighbours farther than the point's own radius dropped.
    seg_off (additive): `points` holds several independent clouds -- one graph whose edges never cross clouds."""
    if points.shape[0] == 0:
        g = Graph(points, torch.zeros((0, 2), dtype=torch.int64, device=points.device),
                  torch.zeros((0,), dtype=torch.float32, device=points.device))
        g.seg_off = seg_off
        return g
    idxs, dists, _ = knn(points, points, K=K, r=-1.0, bound=radii, bound_mode=BOUND_LE, cell=-GRAPH_CELL_DIV,
                         src_seg_off=seg_off, dest_seg_off=seg_off, keep_kernel_width=True)
    return KnnGraph(points, idxs, dists, seg_off)  # the edge list (make_edges) is cut only if somebody reads .edges


@dataclass
class ComponentSet:
    """Connected components with >= minimum_vertices members, size descending (data_types/graph.py:32-51),
    laid out for the per-component kernels: vertices renumbered so each component is contiguous."""
    n_components: int
    comp_size: torch.Tensor  # [C] int32
    comp_off: torch.Tensor  # [C+1] int32
    vert_order: torch.Tensor  # [m] int32 original vertex ids, grouped by component, ascending inside
    new_id: torch.Tensor  # [n] int32, -1 for vertices of dropped components
    labels: torch.Tensor  # [n] int32 smallest member id of each vertex's component
    row_off: torch.Tensor  # CSR over the renumbered vertices (undirected, self loops removed)
    col: torch.Tensor
    wgt: torch.Tensor
    # batched clouds (graph.seg_off): components are ordered cloud by cloud
    n_seg: int = 1
    comp_seg: Optional[torch.Tensor] = None  # [C] int32 cloud of each component
    comp_seg_off: Optional[torch.Tensor] = None  # [n_seg+1] int32 component range of each cloud
    vert_seg_off: Optional[torch.Tensor] = None  # [n_seg+1] int32 the clouds' ranges in the renumbered vertex space

    def __len__(self):
        return self.n_components

    def vertices(self, c: int) -> torch.Tensor:
        a, b = int(self.comp_off[c]), int(self.comp_off[c + 1])
        return self.vert_order[a:b].long()


def connected_components(graph: Graph, minimum_vertices: int = 0) -> ComponentSet:
    L = _lib.lib()
    dev = graph.vertices.device
    n = graph.vertices.shape[0]
    seg_off = getattr(graph, "seg_off", None)
    nseg = 1 if seg_off is None else int(seg_off.shape[0]) - 1
    from_knn = isinstance(graph, KnnGraph) and graph._cap_cache is None  # labels / adjacency straight from the search tables
    i32 = lambda k: torch.empty((max(k, 1),), dtype=torch.int32, device=dev)
    labels = i32(n)
    ws = _lib.workspace(L.st_connected_components_workspace_bytes(n), dev)
    if from_knn:
        idxs, dists = graph.idxs.contiguous(), graph.dists.contiguous()
        K = idxs.shape[1]
        E = n * K
        first_of = None
        if nseg > 1:  # first vertex of every vertex's cloud ("vertex 0" of make_edges' idx > 0 rule)
            cloud_of = torch.bucketize(torch.arange(n, device=dev, dtype=torch.int32), seg_off[1:].contiguous(), right=True)
            first_of = seg_off[cloud_of.clamp(max=nseg - 1)].contiguous()
        _lib.check(L.st_connected_components_knn(_lib.ptr(idxs), n, K, _lib.ptr(first_of), _lib.ptr(labels), _lib.ptr(ws),
                                                 ws.numel(), _lib.stream(dev)))
    else:
        edges, w = graph.padded if isinstance(graph, PaddedGraph) else (graph.edges, graph.edge_weights)
        edges, w = edges.contiguous(), w.contiguous()
        E = edges.shape[0]
        _lib.check(L.st_connected_components(_lib.ptr(edges), E, n, _lib.ptr(labels), _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    comp_size, comp_off, vert_order, new_id = i32(n), i32(n + 1), i32(n), i32(n)
    comp_seg, comp_seg_off, vert_seg_off = (i32(n), i32(nseg + 1), i32(nseg + 1)) if nseg > 1 else (None, None, None)
    nc, nk = ctypes.c_int64(0), ctypes.c_int64(0)
    ws = _lib.workspace(L.st_component_layout_workspace_bytes(n), dev)
    _lib.check(L.st_component_layout_seg(_lib.ptr(labels), n, int(minimum_vertices), _lib.ptr(seg_off) if nseg > 1 else None, nseg,
                                         _lib.ptr(comp_size), _lib.ptr(comp_off), _lib.ptr(vert_order), _lib.ptr(new_id),
                                         _lib.ptr(comp_seg), _lib.ptr(comp_seg_off), _lib.ptr(vert_seg_off),
                                         ctypes.byref(nc), ctypes.byref(nk), _lib.ptr(ws), ws.numel(), _lib.stream(dev),
                                         None))  # (max_comp_host: optional, nobody needs the largest component's size on the host)
    C, m = nc.value, nk.value
    row_off, col, wgt = i32(m + 1), i32(2 * E), torch.empty((max(2 * E, 1),), dtype=torch.float32, device=dev)
    if m > 0:
        ws = _lib.workspace(L.st_component_csr_knn_workspace_bytes(m, n, K) if from_knn else L.st_component_csr_workspace_bytes(m), dev)
        if from_knn:
            _lib.check(L.st_component_csr_knn(_lib.ptr(idxs), _lib.ptr(dists), n, K, _lib.ptr(first_of),
                                              _lib.ptr(new_id), m, _lib.ptr(row_off), _lib.ptr(col), _lib.ptr(wgt), _lib.ptr(ws),
                                              ws.numel(), _lib.stream(dev)))
        else:
            _lib.check(L.st_component_csr(_lib.ptr(edges), _lib.ptr(w), E, _lib.ptr(new_id), m, _lib.ptr(row_off), _lib.ptr(col),
                                          _lib.ptr(wgt), _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    return ComponentSet(C, comp_size[:C], comp_off[: C + 1], vert_order[:m], new_id[:n], labels[:n], row_off[: m + 1], col, wgt,
                        nseg, comp_seg[:C] if comp_seg is not None else None, comp_seg_off, vert_seg_off)


def remap_edges(edges: torch.Tensor) -> torch.Tensor:
    """graph.py:94-104: renumber vertex ids by rank (kept for API parity; the kernels use ComponentSet.new_id)."""
    _, inverse = torch.unique(edges, return_inverse=True)
    return inverse.reshape(edges.shape)
