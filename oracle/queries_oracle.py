"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of `pts_to_nearest_tube_gpu` (smart_tree/util/queries.py:107-133) / `skeleton_to_points` (:139-166):
every point against every tapered tube, float32 with the operation order the HIP kernel uses
(csrc/queries.hip; the same as oracle/pipeline_oracle.nearest_tube_offset): dot = (x*x' + y*y') + z*z',
t = dot(ap, ab) / dot(ab, ab) clipped to [0, 1], NaN (zero-length tube) counts as the minimal score, first minimum.
The reference evaluates the same expression with torch einsums (summation order unspecified): pinned against it by
tests/golden/nearest_tube.npz to 1e-5 (indices equal wherever the two best scores are further apart than that).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def nearest_tube(pts, a, b, r1, r2, chunk: int = 2048):
    """Returns (vectors [n,3] projection - point, idx [n] int64, radius [n])."""
    pts, a, b = np.asarray(pts, F32), np.asarray(a, F32), np.asarray(b, F32)
    r1, r2 = np.asarray(r1, F32).reshape(-1), np.asarray(r2, F32).reshape(-1)
    ab = b - a
    ab2 = _dot(ab, ab)
    vec = np.empty((len(pts), 3), F32)
    idx = np.empty(len(pts), np.int64)
    rad = np.empty(len(pts), F32)
    for s in range(0, len(pts), chunk):
        p = pts[s:s + chunk, None, :]
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (_dot(p - a[None], ab[None]) / ab2[None]).astype(F32)
        t = np.where(t < 0, F32(0), np.where(t > 1, F32(1), t)).astype(F32)
        proj = (a[None] + t[..., None] * ab[None]).astype(F32)
        r = ((F32(1) - t) * r1[None] + t * r2[None]).astype(F32)
        v = (proj - p).astype(F32)
        with np.errstate(invalid="ignore"):
            score = np.abs(np.sqrt(_dot(v, v)).astype(F32) - r)
        nan = np.isnan(score)
        best = np.where(nan.any(1), nan.argmax(1), np.where(nan, np.inf, score).argmin(1))
        rows = np.arange(len(best))
        vec[s:s + chunk], idx[s:s + chunk], rad[s:s + chunk] = v[rows, best], best, r[rows, best]
    return vec, idx, rad
