"""ORACLE (test infrastructure only -- never imported by the product path).

ctypes front-end of oracle/skeleton_oracle.c plus the numpy glue that restates
`Skeletonizer.forward` / `process_subgraph` (reference skeleton/skeletonize.py:31-95) and the
host-side post-processing inputs.  See the C file's header for the reference citations, the
third-party semantics that had to be restated, and the canonical tie-break choices.
"""
from __future__ import annotations

import ctypes
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import List

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libskeleton_oracle.so"
_lib = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i64 = ctypes.c_int64


def build() -> Path:
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        L = ctypes.CDLL(str(_SO))
        L.so_knn.argtypes = [_i64, _f32p, _i64, _f32p, ctypes.c_int, ctypes.c_float, _i64p, _f32p]
        L.so_knn.restype = None
        L.so_outlier_mask.argtypes = [_i64, _f32p, _f32p, ctypes.c_int, _u8p]
        L.so_outlier_mask.restype = None
        L.so_nn_graph.argtypes = [_i64, _f32p, _f32p, ctypes.c_int, _i64p, _f32p]
        L.so_nn_graph.restype = _i64
        L.so_cc_labels.argtypes = [_i64, _i64, _i64p, _i64p]
        L.so_cc_labels.restype = None
        L.so_sssp.argtypes = [_i64, _i64, _i64p, _f32p, _i64, _f32p, _i64p]
        L.so_sssp.restype = None
        L.so_tree_distance.argtypes = [_i64, _f32p, _i64p, _i64, _f32p, _f32p]
        L.so_tree_distance.restype = None
        L.so_sample_tree.argtypes = [_i64, _f32p, _f32p, _i64p, _f32p, _i64p, _i64p, _i64p, _i64p, _i64p, _i64p]
        L.so_sample_tree.restype = _i64
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64a(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------------------ primitives ---
def knn(src, dst, K: int, r: float):
    """graph.py:12-26: idx [n1,K] (-1 pad), dist = sqrt(FRNN d2) (NaN where padded)."""
    src, dst = _f32(src), _f32(dst)
    idx = np.empty((len(src), K), np.int64)
    d2 = np.empty((len(src), K), np.float32)
    lib().so_knn(len(src), src, len(dst), dst, K, np.float32(r), idx, d2)
    with np.errstate(invalid="ignore"):
        return idx, np.sqrt(d2)


def outlier_removal(points, radii, nb_points: int = 8):
    points, radii = _f32(points), _f32(radii).reshape(-1)
    keep = np.zeros(len(points), np.uint8)
    lib().so_outlier_mask(len(points), points, radii, nb_points, keep)
    return keep.astype(bool)


def nn_graph(points, radii, K: int = 16):
    """radii must already be clamped (skeletonize.py:39).  Returns edges [E,2] int64, weights [E]."""
    points, radii = _f32(points), _f32(radii).reshape(-1)
    n = len(points)
    edges = np.empty((max(n * K, 1), 2), np.int64)
    w = np.empty(max(n * K, 1), np.float32)
    E = lib().so_nn_graph(n, points, radii, K, edges, w)
    return edges[:E].copy(), w[:E].copy()


def cc_labels(n: int, edges):
    edges = _i64a(edges).reshape(-1, 2)
    label = np.empty(n, np.int64)
    lib().so_cc_labels(n, len(edges), edges, label)
    return label


def sssp(n: int, edges, w, root: int):
    edges, w = _i64a(edges).reshape(-1, 2), _f32(w)
    dist = np.empty(n, np.float32)
    pred = np.empty(n, np.int64)
    lib().so_sssp(n, len(edges), edges, w, root, dist, pred)
    return dist, pred


def tree_distance(points, pred, root: int):
    points, pred = _f32(points), _i64a(pred)
    out = np.empty(len(points), np.float32)
    lib().so_tree_distance(len(points), points, pred, root, out, out)
    return out


@dataclass
class OracleBranch:
    branch_id: int
    parent_id: int
    verts: np.ndarray  # component-local vertex indices, root side first


def sample_tree(points, radii, preds, distances):
    points, radii = _f32(points), _f32(radii).reshape(-1)
    preds, distances = _i64a(preds), _f32(distances)
    n = len(points)
    cap = max(n, 1)
    parent = np.empty(cap, np.int64)
    off = np.empty(cap, np.int64)
    ln = np.empty(cap, np.int64)
    verts = np.empty(cap, np.int64)
    bop = np.empty(cap, np.int64)
    iters = np.zeros(1, np.int64)
    nb = lib().so_sample_tree(n, points, radii, preds, distances, parent, off, ln, verts, bop, iters)
    branches = [OracleBranch(b, int(parent[b]), verts[off[b]:off[b] + ln[b]].copy()) for b in range(nb)]
    return branches, bop[:n].copy(), int(iters[0])


# ------------------------------------------------------------------------- whole stage glue ---
@dataclass
class OracleComponent:
    vertex_ids: np.ndarray  # indices into the outlier-filtered cloud, ascending
    root: int  # local index
    preds: np.ndarray  # local, -1 at the root
    dist: np.ndarray  # first SSSP
    tree_dist: np.ndarray  # second SSSP on the predecessor tree
    branches: List[OracleBranch] = field(default_factory=list)
    branch_of_point: np.ndarray = None


@dataclass
class OracleSkeleton:
    keep_mask: np.ndarray  # outlier_removal over the input cloud
    edges: np.ndarray
    weights: np.ndarray
    labels: np.ndarray
    components: List[OracleComponent]


def skeletonize(xyz, medial_vector, K: int = 16, min_connection_length: float = 0.02,
                minimum_graph_vertices: int = 32) -> OracleSkeleton:
    """Skeletonizer.forward (skeletonize.py:31-55) on one branch cloud."""
    xyz, mv = _f32(xyz), _f32(medial_vector)
    medial = xyz + mv  # cloud.py:229-231
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)  # cloud.py:254-256
    keep = outlier_removal(medial, radius, 8)
    xyz, medial, radius = xyz[keep], medial[keep], radius[keep]
    n = len(xyz)
    edges, w = nn_graph(medial, np.maximum(radius, np.float32(min_connection_length)), K)
    labels = cc_labels(n, edges) if n else np.zeros(0, np.int64)
    comps = []
    if n:
        roots, counts = np.unique(labels, return_counts=True)
        order = np.lexsort((roots, -counts))  # size desc, then smallest member asc
        for ci in order:
            if counts[ci] < minimum_graph_vertices:
                continue
            ids = np.nonzero(labels == roots[ci])[0]
            comps.append(_process_component(ids, xyz, medial, radius, edges, w))
    return OracleSkeleton(keep, edges, w, labels, comps)


def _process_component(ids, xyz, medial, radius, edges, w) -> OracleComponent:
    """process_subgraph (skeletonize.py:57-95)."""
    inside = np.isin(edges[:, 0], ids)  # a component is edge-closed: src inside <=> dst inside
    local = np.searchsorted(ids, edges[inside])  # remap_edges, graph.py:94-104
    lw = w[inside]
    pts, rad = medial[ids], radius[ids]
    root = int(np.argmin(xyz[ids, 1]))  # cloud.py:204-206 (first minimum)
    dist, pred = sssp(len(ids), local, lw, root)
    tdist = tree_distance(pts, pred, root)
    branches, bop, _ = sample_tree(pts, rad, pred, tdist)
    return OracleComponent(ids, root, pred, dist, tdist, branches, bop)
