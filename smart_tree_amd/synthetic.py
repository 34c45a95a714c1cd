"""Seeded procedural tree clouds (SURVEY.md section 8d "Synthetic inputs").

The reference ships no data (its dataset is an external download), so BASELINE.json's
configurations are defined on this generator: a recursive tapered-cylinder tree, points sampled
proportionally to lateral surface area with 2 mm Gaussian noise, exact medial vectors as ground
truth, optional Gaussian foliage blobs at the twig tips.  Pure numpy, deterministic per seed.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class TreeSegments:
    a: np.ndarray  # [S,3] start of each tapered cylinder
    b: np.ndarray  # [S,3] end
    ra: np.ndarray  # [S] start radius
    rb: np.ndarray  # [S] end radius
    depth: np.ndarray  # [S] recursion depth
    is_tip: np.ndarray  # [S] bool, segment has no children


def _orthobasis(d: np.ndarray):
    """Two unit vectors orthogonal to each row of d (rows unit length)."""
    helper = np.where(np.abs(d[:, [1]]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    u = np.cross(d, helper)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(d, u)
    return u, v


def grow_tree(seed: int = 0, scale: float = 1.0, max_depth: int = 7) -> TreeSegments:
    rng = np.random.RandomState(seed)
    segs = []  # (a, b, ra, rb, depth, is_tip)

    def rec(a, direction, length, radius, depth):
        b = a + direction * length
        n_child = 0 if depth >= max_depth else int(rng.randint(2, 4))
        r_end = radius * 0.75
        segs.append([a, b, radius, r_end, depth, n_child == 0])
        for _ in range(n_child):
            # child direction: tilt 20-50 degrees off the parent, random azimuth, slight upward bias
            tilt = np.deg2rad(rng.uniform(20.0, 50.0))
            az = rng.uniform(0.0, 2.0 * np.pi)
            u, v = _orthobasis(direction[None, :])
            nd = np.cos(tilt) * direction + np.sin(tilt) * (np.cos(az) * u[0] + np.sin(az) * v[0])
            nd = nd + np.array([0.0, 0.15, 0.0])
            nd /= np.linalg.norm(nd)
            rec(b, nd, length * rng.uniform(0.6, 0.85), r_end * rng.uniform(0.6, 0.9), depth + 1)

    rec(np.zeros(3), np.array([0.0, 1.0, 0.0]), 3.0 * scale, 0.20 * scale, 0)
    a = np.array([s[0] for s in segs])
    b = np.array([s[1] for s in segs])
    return TreeSegments(
        a=a,
        b=b,
        ra=np.array([s[2] for s in segs]),
        rb=np.array([s[3] for s in segs]),
        depth=np.array([s[4] for s in segs]),
        is_tip=np.array([s[5] for s in segs], dtype=bool),
    )


def sample_tree_cloud(
    n_points: int,
    seed: int = 0,
    scale: float = 1.0,
    noise: float = 0.002,
    foliage_fraction: float = 0.0,
    max_depth: int = 7,
):
    """Returns dict(xyz, rgb, medial_vector, class_l) as float32 numpy arrays.

    class_l: 0 = branch, 1 = foliage.  medial_vector of a branch point is -r * u_hat (the vector
    from the un-noised surface point to its axis point); foliage gets a zero vector.
    """
    tree = grow_tree(seed, scale, max_depth)
    rng = np.random.RandomState(seed + 7919)

    n_fol = int(round(n_points * foliage_fraction))
    n_br = n_points - n_fol

    axis = tree.b - tree.a
    length = np.linalg.norm(axis, axis=1)
    d = axis / length[:, None]
    area = np.pi * (tree.ra + tree.rb) * length
    seg = rng.choice(len(area), size=n_br, p=area / area.sum())
    t = rng.uniform(0.0, 1.0, n_br)
    theta = rng.uniform(0.0, 2.0 * np.pi, n_br)
    u, v = _orthobasis(d)
    radial = np.cos(theta)[:, None] * u[seg] + np.sin(theta)[:, None] * v[seg]
    r = tree.ra[seg] * (1.0 - t) + tree.rb[seg] * t
    axis_pt = tree.a[seg] + t[:, None] * axis[seg]
    xyz = axis_pt + r[:, None] * radial + rng.normal(0.0, noise, (n_br, 3))
    mv = -r[:, None] * radial
    cls = np.zeros((n_br, 1))

    if n_fol > 0:
        tips = tree.b[tree.is_tip]
        which = rng.randint(0, len(tips), n_fol)
        fxyz = tips[which] + rng.normal(0.0, 0.08 * scale, (n_fol, 3))
        xyz = np.concatenate([xyz, fxyz])
        mv = np.concatenate([mv, np.zeros((n_fol, 3))])
        cls = np.concatenate([cls, np.ones((n_fol, 1))])
        perm = rng.permutation(n_points)
        xyz, mv, cls = xyz[perm], mv[perm], cls[perm]

    return {
        "xyz": xyz.astype(np.float32),
        "rgb": np.zeros((n_points, 3), dtype=np.float32),
        "medial_vector": mv.astype(np.float32),
        "class_l": cls.astype(np.float32),
    }
