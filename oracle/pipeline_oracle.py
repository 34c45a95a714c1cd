"""ORACLE (test infrastructure only -- never imported by the product path).

`Pipeline.process_cloud` (reference smart_tree/pipeline.py:55-106) restated on the CPU by chaining
the stage oracles, plus a numpy restatement of the host-side post-processing
(smart_tree/data_types/tree.py:73-134,164-176; util/queries.py:89-133; data_types/branch.py:61-67).
It is also the `cpu_baseline` leg of bench.py ("port": the reference itself has no CPU path --
spconv-cu117, FRNN and cugraph are CUDA-only).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch

from . import skeleton_oracle as so
from . import unet_oracle as uo
from . import voxel_oracle as vo

F32 = np.float32


@dataclass
class OBranch:
    _id: int
    parent_id: int
    xyz: np.ndarray  # [m,3] float32
    radii: np.ndarray  # [m,1] float32 ([m] after smooth -- quirk kept, tree.py:130-134)


@dataclass
class OTree:
    _id: int
    branches: Dict[int, OBranch] = field(default_factory=dict)


# ------------------------------------------------------------------------- post-processing ---
# Float32 operation order (shared with csrc/postprocess.hip; the reference leaves it to torch's
# einsum / norm / sum / conv1d kernels, which differ from this only in the last bit):
#   dot(a,b) = (ax*bx + ay*by) + az*bz;  length = sequential sum of segment norms;
#   box filter = sequential sum over the window of r*w with w = fl32(1/k).
def _dot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def branch_length(b: OBranch) -> F32:
    """branch.py:61-63."""
    d = (b.xyz[1:] - b.xyz[:-1]).astype(F32)
    seg = np.sqrt(_dot(d, d)).astype(F32)
    return np.cumsum(seg, dtype=F32)[-1] if len(seg) else F32(0)


def prune(tree: OTree, min_radius: float, min_length: float) -> None:
    """tree.py:94-121 (thresholds compared in float32, as torch compares a float32 tensor with a Python float)."""
    root_id = min(tree.branches.keys())
    keep = {root_id: tree.branches[root_id]}
    for key, b in tree.branches.items():
        orphan = b.parent_id not in keep and b._id != root_id
        initial_radius = max(b.radii.reshape(-1)[0], b.radii.reshape(-1)[-1])  # branch.py:65-67
        if not (orphan or branch_length(b) < F32(min_length) or initial_radius < F32(min_radius)):
            keep[key] = b
    tree.branches = keep


def nearest_tube_offset(pt: np.ndarray, parent: OBranch) -> np.ndarray:
    """queries.py:89-133 for one point against the parent's tube chain: vector to the projection on the
    tube that minimises |distance - interpolated radius| (first minimum; NaN counts as minimal)."""
    p = pt.reshape(1, 3).astype(F32)
    a, b = parent.xyz[:-1].astype(F32), parent.xyz[1:].astype(F32)
    r1, r2 = parent.radii.reshape(-1)[:-1].astype(F32), parent.radii.reshape(-1)[1:].astype(F32)
    ab = b - a
    ap = p - a
    with np.errstate(invalid="ignore", divide="ignore"):
        t = _dot(ap, ab) / _dot(ab, ab)
    t = np.where(t < 0, F32(0), np.where(t > 1, F32(1), t)).astype(F32)
    proj = a + t[:, None] * ab
    r = (F32(1) - t) * r1 + t * r2
    d = proj - p
    score = np.abs(np.sqrt(_dot(d, d)) - r)
    i = int(np.argmax(np.isnan(score))) if np.isnan(score).any() else int(np.argmin(score))
    return proj[i] - p[0]


def repair(tree: OTree) -> None:
    """tree.py:73-92."""
    ids = set(b._id for b in tree.branches.values())
    for b in tree.branches.values():
        if b.parent_id not in ids:
            continue
        parent = tree.branches[b.parent_id]
        v = nearest_tube_offset(b.xyz[0], parent)
        b.xyz = np.concatenate([(b.xyz[0] + v).reshape(1, 3), b.xyz]).astype(F32)  # tree.py:89-91
        b.radii = np.concatenate([b.radii[[0]], b.radii])


def smooth(tree: OTree, kernel_size: int) -> None:
    """tree.py:123-134: zero-padded box filter (F.conv1d padding="same"), only when len > kernel; radii become 1-D."""
    w = F32(1) / F32(kernel_size)
    half = (kernel_size - 1) // 2  # torch pads left (k-1)//2, right k-1-left (even kernels: the extra sample goes right)
    for b in tree.branches.values():
        n = b.radii.shape[0]
        if n > kernel_size:
            padded = np.zeros(n + kernel_size, F32)
            padded[half: half + n] = b.radii.reshape(-1)
            acc = np.zeros(n, F32)
            for j in range(kernel_size):
                acc = acc + padded[j: j + n] * w
            b.radii = acc


def post_process(trees: List[OTree], prune_skeletons=True, min_radius=0.01, min_length=0.02, repair_skeletons=True,
                 smooth_skeletons=True, smooth_kernel_size=11) -> None:
    """pipeline.py:95-106 + DisjointTreeSkeleton (tree.py:164-176): only skeleton 0 is pruned."""
    if prune_skeletons and trees:
        prune(trees[0], min_radius, min_length)
    if repair_skeletons:
        for t in trees:
            repair(t)
    if smooth_skeletons:
        for t in trees:
            smooth(t, smooth_kernel_size)


# ----------------------------------------------------------------------------- whole path ---
def labelled_cloud(xyz, rgb, weights, voxel_size, block_size=4.0, buffer_size=0.4, dtype=torch.float32, timings=None):
    """CentreCloud + ModelInference.forward (model_inference.py:49-100).  Returns dict(xyz, rgb, medial_vector, class_l)."""
    t0 = time.perf_counter()
    xyz = vo.centre_cloud(xyz)
    vx = vo.voxelize_cloud(xyz, rgb, voxel_size, block_size, buffer_size)
    t1 = time.perf_counter()
    out = uo.OracleNet(weights, dtype=dtype).forward(vx["feats"][:, :3], vx["coords"])
    mv, cls = uo.inference_tail(out["radius"], out["direction"], out["class_l"])
    t2 = time.perf_counter()
    if timings is not None:
        timings["voxelize"] = t1 - t0
        timings["unet"] = t2 - t1
    m = vx["mask"]
    return {"xyz": vx["feats"][m, :3], "rgb": vx["feats"][m, 3:6], "medial_vector": mv[m].astype(F32), "class_l": cls[m],
            "n_voxels": len(m)}


def skeleton_from_labelled(xyz, medial_vector, class_l, branch_classes=(0,), K=16, min_connection_length=0.02,
                           minimum_graph_vertices=32) -> List[OTree]:
    """filter_by_class + Skeletonizer.forward (pipeline.py:67-71), branches as path.py:128-133 builds them."""
    sel = np.isin(class_l.reshape(-1), np.asarray(branch_classes))
    xyz, mv = xyz[sel].astype(F32), medial_vector[sel].astype(F32)
    sk = so.skeletonize(xyz, mv, K, min_connection_length, minimum_graph_vertices)
    kept = np.nonzero(sk.keep_mask)[0]
    medial = (xyz + mv)[kept]
    radius = np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(F32)[kept]
    trees = []
    for c, comp in enumerate(sk.components):
        t = OTree(c)
        for b in comp.branches:
            ids = comp.vertex_ids[b.verts]
            t.branches[b.branch_id] = OBranch(b.branch_id, b.parent_id, medial[ids], radius[ids].reshape(-1, 1))
        trees.append(t)
    return trees


def process_cloud(xyz, rgb, weights, voxel_size, timings=None, **post_kw) -> List[OTree]:
    lc = labelled_cloud(xyz, rgb, weights, voxel_size, timings=timings)
    t0 = time.perf_counter()
    trees = skeleton_from_labelled(lc["xyz"], lc["medial_vector"], lc["class_l"])
    t1 = time.perf_counter()
    post_process(trees, **post_kw)
    if timings is not None:
        timings["skeleton"] = t1 - t0
        timings["post_process"] = time.perf_counter() - t1
    return trees
