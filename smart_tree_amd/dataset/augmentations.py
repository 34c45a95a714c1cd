"""Preprocessing used by the inference pipeline (reference smart_tree/dataset/augmentations.py).

Only `CentreCloud` (:38-41) and `AugmentationPipeline` (:108-116) are on the inference path; the
training-time augmentations are out of scope."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Sequence

import torch

from .. import _lib
from ..data_types.cloud import Cloud


class Augmentation(ABC):
    @abstractmethod
    def __call__(self, cloud: Cloud) -> Cloud:
        ...


class CentreCloud(Augmentation):
    """x and z centred on the bounding box, lowest point moved to y = 0; drops every field but xyz/rgb
    (Cloud.translate), as the reference does."""

    def __call__(self, cloud: Cloud) -> Cloud:
        if cloud.xyz.is_cuda or _lib._ALLOW_HOST_POINTERS:
            # one bounding-box pass + one translate pass (csrc/graph.hip st_centre_cloud), same float32 arithmetic; a batch
            # (Cloud.collate) centres every cloud on its own box in the same two launches
            L = _lib.lib()
            xyz = cloud.xyz.contiguous().float()
            out = torch.empty_like(xyz)
            nseg = cloud.n_seg
            ws = _lib.workspace(256 + 24 * nseg, xyz.device)
            _lib.check(L.st_centre_cloud_seg(_lib.ptr(xyz), xyz.shape[0], _lib.ptr(cloud.seg_off), nseg, _lib.ptr(out), _lib.ptr(ws),
                                             ws.numel(), _lib.stream(xyz.device)))
            return Cloud(out, cloud.rgb, seg_off=cloud.seg_off)
        if cloud.seg_off is not None:
            raise _lib.StError("batched clouds are centred on the GPU only")
        centre, half = cloud.bbox  # host tensors (e.g. before upload): plain torch, as the reference writes it
        lift = torch.zeros(3, device=centre.device, dtype=centre.dtype)
        lift[1] = half[1]
        return cloud.translate(-centre + lift)


class AugmentationPipeline(Augmentation):
    def __init__(self, augmentations: Sequence[Augmentation]):
        self.augmentations = list(augmentations)

    def __call__(self, cloud: Cloud) -> Cloud:
        for aug in self.augmentations:
            cloud = aug(cloud)
        return cloud
