"""Tube meshes of skeletons without open3d (reference smart_tree/o3d_abstractions/geometries.py:157-211 over
util/maths.py:158-186): every branch becomes a tube of `n`-gons around its vertices, radii from the branch, the rings kept
untwisted by parallel-transporting a tangent along the branch; `Pipeline(save_outputs=True)` writes the merged mesh as
`mesh.ply` like the reference (pipeline.py:90).  Host code on small arrays -- not part of the GPU hot path."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np


def unit_circle(n: int) -> np.ndarray:
    a = np.linspace(0, 2 * np.pi, n + 1)[:-1]
    return np.stack([np.sin(a), np.cos(a)], axis=1)


def cylinder_triangles(m: int, n: int) -> np.ndarray:
    """Triangles of n rings of m vertices each (ring r = vertices [r*m, (r+1)*m)): two per quad, all first halves then all
    second halves -- the reference's order (geometries.py:161-177)."""
    v0 = np.arange(m)
    v1 = (v0 + 1) % m
    quads = np.stack([v0, v1, v1 + m, v0 + m], axis=1)
    quads = (quads[None] + (np.arange(n - 1) * m).reshape(n - 1, 1, 1)).reshape(-1, 4)
    return np.concatenate([quads[:, [0, 1, 2]], quads[:, [2, 3, 0]]])


def vertex_dirs(points: np.ndarray) -> np.ndarray:
    """Unit direction at every vertex: first / last segment at the ends, mean of the adjacent segments inside
    (maths.py:158-165; the reference scales all segments by one common norm first, which cancels in the result)."""
    d = points[1:] - points[:-1]
    d = d / np.linalg.norm(d)
    smooth = (d[1:] + d[:-1]) * 0.5
    dirs = np.concatenate([d[0:1], smooth, d[-2:-1]])
    return dirs / np.linalg.norm(dirs, axis=1, keepdims=True)


def gen_tangents(dirs: np.ndarray, t: np.ndarray) -> np.ndarray:
    """Parallel transport (maths.py:173-186): t_i = (d_i x t_{i-1}) normalised, crossed with d_i again."""
    out = []
    for d in dirs:
        c = np.cross(d, t)
        c = c / np.linalg.norm(c, axis=-1, keepdims=True)
        t = np.cross(c, d)
        out.append(t)
    return np.stack(out)


def tube_vertices(points: np.ndarray, radii: np.ndarray, n: int = 10, start: Optional[np.ndarray] = None) -> np.ndarray:
    """[len(points), n, 3] ring vertices (geometries.py:180-189).  `start`: the vector the first tangent is derived from
    (the reference draws a random unit vector; pass one for reproducible meshes)."""
    points = np.asarray(points, dtype=np.float32)
    circle = unit_circle(n).astype(np.float32)
    dirs = vertex_dirs(points)
    if start is None:
        x = np.random.randn(3).astype(np.float32)
        start = x / np.linalg.norm(x)
    t = gen_tangents(dirs, np.asarray(start, dtype=np.float32))
    b = np.stack([t, np.cross(t, dirs)], axis=1) * np.asarray(radii, dtype=np.float32).reshape(-1, 1, 1)
    return np.einsum("bdx,md->bmx", b, circle) + points.reshape(points.shape[0], 1, 3)


def tube_mesh(points, radii, n: int = 10, start=None) -> Tuple[np.ndarray, np.ndarray]:
    """(vertices [len*n, 3] float32, triangles [T, 3] int64) of one branch (geometries.py:202-211)."""
    v = tube_vertices(points, radii, n, start)
    rings, m, _ = v.shape
    return v.reshape(-1, 3).astype(np.float32), cylinder_triangles(m, rings)


def skeleton_mesh(skeleton, n: int = 10, start=None) -> Tuple[np.ndarray, np.ndarray]:
    """All branches of a TreeSkeleton / DisjointTreeSkeleton merged into one mesh (tree.py:40-44, o3d_merge_meshes)."""
    trees = skeleton.skeletons if hasattr(skeleton, "skeletons") else [skeleton]
    verts, tris, base = [], [], 0
    for tree in trees:
        for b in tree.branches.values():
            if len(b) < 2:
                continue
            v, t = tube_mesh(b.xyz.numpy(), b.radii.reshape(-1).numpy(), n, start)
            verts.append(v)
            tris.append(t + base)
            base += len(v)
    if not verts:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    return np.concatenate(verts), np.concatenate(tris)


def write_ply_mesh(path, vertices: np.ndarray, triangles: np.ndarray, colour=(1.0, 0.0, 0.0)) -> None:
    """Binary little-endian PLY triangle mesh with one uniform vertex colour (what save_o3d_mesh writes for a painted mesh)."""
    vrec = np.empty(len(vertices), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    vrec["x"], vrec["y"], vrec["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    c = np.clip(np.asarray(colour, dtype=np.float32) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    vrec["red"], vrec["green"], vrec["blue"] = c[0], c[1], c[2]
    frec = np.empty(len(triangles), dtype=[("n", "u1"), ("a", "<i4"), ("b", "<i4"), ("c", "<i4")])
    frec["n"] = 3
    frec["a"], frec["b"], frec["c"] = triangles[:, 0], triangles[:, 1], triangles[:, 2]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(vertices)}", "property float x", "property float y",
              "property float z", "property uchar red", "property uchar green", "property uchar blue",
              f"element face {len(triangles)}", "property list uchar int vertex_indices", "end_header"]
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(vrec.tobytes())
        f.write(frec.tobytes())
