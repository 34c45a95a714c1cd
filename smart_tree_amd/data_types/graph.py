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
