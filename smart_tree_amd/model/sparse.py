"""Sparse tensor container + batch helpers (reference smart_tree/model/sparse.py:9-61).

`SparseConvTensor` stands in for spconv's class with the attributes the reference touches
(`features`, `indices`, `spatial_shape`, `batch_size`, `replace_feature`; uses at
model_blocks.py:149-155,227-240 and sparse.py:17-19).
"""
from __future__ import annotations

from typing import List

import torch


class SparseConvTensor:
    def __init__(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int):
        self.features = features
        self.indices = indices  # [N,4] int32 (batch, z, y, x)
        self._spatial_shape = spatial_shape  # None = not reduced yet (see the property)
        self.batch_size = batch_size
        self.indice_dict = {}
        self.blk_seg = None  # batched clouds (Cloud.collate): cloud of every batch index coords[:,0] -> per-cloud spatial extents
        self.n_seg = 1
        # optional host-side bounds (number of batch indices, exclusive bound of z / y / x) from whoever made the voxels:
        # lets the network build its rulebooks from occupancy bricks (csrc/brick.hip) instead of hash tables
        self.brick_hint = None

    @property
    def spatial_shape(self):
        """The reference's attribute (sparse.py:15-18: the largest z / y / x, not +1).  No kernel of this package reads it
        (they take the extent from the indices), so the reduction over the coordinates runs when somebody asks."""
        if self._spatial_shape is None:
            c = self.indices
            self._spatial_shape = torch.max(c, 0)[0][1:] if c.shape[0] else torch.zeros(3, dtype=c.dtype, device=c.device)
        return self._spatial_shape

    @spatial_shape.setter
    def spatial_shape(self, value):
        self._spatial_shape = value

    def replace_feature(self, new_features: torch.Tensor) -> "SparseConvTensor":
        out = SparseConvTensor(new_features, self.indices, self._spatial_shape, self.batch_size)
        out.indice_dict = self.indice_dict
        out.blk_seg, out.n_seg, out.brick_hint = self.blk_seg, self.n_seg, self.brick_hint
        return out


def sparse_from_batch(features: torch.Tensor, coordinates: torch.Tensor, device, blk_seg: torch.Tensor = None,
                      n_seg: int = 1, brick_hint=None) -> SparseConvTensor:
    """Reference sparse.py:9-19, quirks kept on the attributes: spatial_shape = max coordinate (not
    +1) and batch_size = number of voxels.  The kernels derive the true extent from the indices."""
    batch_size = features.shape[0]
    features = features.to(device)
    coordinates = coordinates.to(device)
    out = SparseConvTensor(features.contiguous(), coordinates.int().contiguous(), None, batch_size=batch_size)
    if blk_seg is not None and n_seg > 1:
        out.blk_seg, out.n_seg = blk_seg.to(device).int().contiguous(), n_seg
    out.brick_hint = brick_hint
    return out


def batch_collate(batch):
    """Reference sparse.py:40-61: write the sample index into coords[:,0], concatenate.  Items whose features are an
    (input, target) pair -- TreeDataset -- come back as ((inputs, targets), coords, mask, names), the inference form as
    (feats, coords, mask, names)."""
    feats, coords, masks, names = zip(*batch)
    coords = [c.clone() for c in coords]
    for i, c in enumerate(coords):
        c[:, 0] = i
    if isinstance(feats[0], tuple):
        inputs, targets = zip(*feats)
        return [(torch.cat(inputs), torch.cat(targets)), torch.cat(coords), torch.cat(masks), names]
    return [torch.cat(feats), torch.cat(coords), torch.cat(masks), names]


def split_sparse(sparse_tensor: SparseConvTensor) -> List:
    ids = sparse_tensor.indices[:, 0]
    n = int(ids.max().item()) + 1 if ids.numel() else 0
    return [(sparse_tensor.indices[ids == i], sparse_tensor.features[ids == i]) for i in range(n)]
