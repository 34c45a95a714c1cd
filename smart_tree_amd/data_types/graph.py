"""Vertex/edge container of the neighbourhood graph.

Reference: `smart_tree/data_types/graph.py:15-51`.  There `connected_cugraph_components`
returns a list of cugraph sub-graphs found through a host loop over every label; here it returns
a `ComponentSet` (device arrays, built by `st_connected_components`) that the skeletonizer
consumes without leaving the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class Graph:
    vertices: torch.Tensor  # [N,3]
    edges: torch.Tensor  # [E,2] int64 (src, dst)
    edge_weights: torch.Tensor  # [E] float32

    def to_device(self, device) -> "Graph":
        return Graph(self.vertices.to(device), self.edges.to(device), self.edge_weights.to(device))

    def connected_cugraph_components(self, minimum_vertices: int = 10):
        # Name kept for drop-in parity with the reference call site (skeletonize.py:43-45).
        from ..skeleton.graph import connected_components

        return connected_components(self, minimum_vertices)

    connected_components = connected_cugraph_components


class PaddedGraph(Graph):
    """What `nn_graph` returns: the edge list at its capacity `[n*K]`, the unused tail padded with (0, 0) self loops of
    weight 0 (a real edge has dst > 0, reference skeleton/graph.py:59), so that building it needs no read-back of the
    edge count.  `connected_cugraph_components` consumes the padded arrays as they are (the padding neither joins
    components nor enters the CSR); `edges` / `edge_weights` -- the reference's `[E,2]` / `[E]` views -- are cut on first
    access, which is where the count comes to the host."""

    def __init__(self, vertices: torch.Tensor, edges_cap: torch.Tensor, weights_cap: torch.Tensor):
        self.vertices = vertices
        self._cap = (edges_cap, weights_cap)
        self._cut = None

    def _trim(self):
        if self._cut is None:
            e, w = self._cap
            E = int((e[:, 1] > 0).sum().item())  # real edges are a prefix of the list
            self._cut = (e[:E], w[:E])
        return self._cut

    @property
    def padded(self):
        return self._cap

    @property
    def edges(self) -> torch.Tensor:
        return self._trim()[0]

    @property
    def edge_weights(self) -> torch.Tensor:
        return self._trim()[1]

    def __repr__(self):
        return f"PaddedGraph(vertices={tuple(self.vertices.shape)}, capacity={self._cap[0].shape[0]})"



class KnnGraph(PaddedGraph):
    """What `nn_graph` returns since round 2: the neighbour search's own tables `idxs` / `dists` [n,K] (after the radius
    filter).  The reference's edge list -- (i, idx) for idx > vertex 0 of i's cloud, graph.py:52-60 -- is the same
    information; `connected_cugraph_components` builds labels and adjacency from the tables directly, and `edges` /
    `edge_weights` / `padded` materialise the int64 edge list only if somebody asks for it."""

    def __init__(self, vertices: torch.Tensor, idxs: torch.Tensor, dists: torch.Tensor, seg_off=None):
        self.vertices = vertices
        self.idxs, self.dists = idxs, dists
        self.seg_off = seg_off
        self._cap_cache = None
        self._cut = None

    @property
    def _cap(self):
        if self._cap_cache is None:
            from ..skeleton.graph import make_edges

            self._cap_cache = make_edges(self.dists, self.idxs, padded=True, seg_off=self.seg_off)
        return self._cap_cache

    def __repr__(self):
        return f"KnnGraph(vertices={tuple(self.vertices.shape)}, K={self.idxs.shape[1]})"
