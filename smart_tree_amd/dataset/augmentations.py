"""Preprocessing used by the inference pipeline (reference smart_tree/dataset/augmentations.py).

`CentreCloud` (:38-41) and `AugmentationPipeline` (:108-116) are on the inference path (CentreCloud runs as one HIP pass);
the training / evaluation-side augmentations (:17-105) are a few tensor expressions on whatever device
the cloud lives on, with the reference's use of torch's global random generator (same seed, same device => same draws)."""
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


class Scale(Augmentation):
    """augmentations.py:17-24: one uniform factor in [min_scale, max_scale) (labels are dropped: Cloud.scale keeps xyz / rgb)."""

    def __init__(self, min_scale=0.9, max_scale=1.1):
        self.min_scale, self.max_scale = min_scale, max_scale

    def __call__(self, cloud: Cloud) -> Cloud:
        t = torch.rand(1, device=cloud.xyz.device) * (self.max_scale - self.min_scale)
        return cloud.scale(t + self.min_scale)


def euler_angles_to_rotation(xyz: torch.Tensor) -> torch.Tensor:
    """util/maths.py:19-46: R_z(z) R_y(y) R_x(x) for angles in radians."""
    x, y, z = (torch.as_tensor(a, dtype=torch.float32).cpu() for a in xyz)
    cx, sx, cy, sy, cz, sz = torch.cos(x), torch.sin(x), torch.cos(y), torch.sin(y), torch.cos(z), torch.sin(z)
    rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
    ry = torch.tensor([[cy, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy]])
    rz = torch.tensor([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    return rz @ (ry @ rx)


class FixedRotate(Augmentation):
    """augmentations.py:27-35 (points are row vectors: xyz @ R)."""

    def __init__(self, xyz):
        self.xyz = xyz

    def __call__(self, cloud: Cloud) -> Cloud:
        self.rot_mat = euler_angles_to_rotation(torch.tensor(self.xyz)).float()
        return cloud.rotate(self.rot_mat)


class VoxelDownsample(Augmentation):
    """augmentations.py:44-49."""

    def __init__(self, voxel_size):
        self.voxel_size = voxel_size

    def __call__(self, cloud: Cloud) -> Cloud:
        return cloud.voxel_down_sample(self.voxel_size)


class FixedTranslate(Augmentation):
    """augmentations.py:52-57."""

    def __init__(self, xyz):
        self.xyz = torch.tensor(xyz)

    def __call__(self, cloud: Cloud) -> Cloud:
        return cloud.translate(self.xyz)


class RandomCrop(Augmentation):
    """augmentations.py:60-72: keep the points that stay inside the bounding box after a random shift."""

    def __init__(self, max_x, max_y, max_z):
        self.max_translation = torch.tensor([max_x, max_y, max_z])

    def __call__(self, cloud: Cloud) -> Cloud:
        dev = cloud.xyz.device
        offset = (torch.rand(3, device=dev) - 0.5) * self.max_translation.to(device=dev)
        p = cloud.xyz + offset
        return cloud.filter(torch.logical_and(p >= cloud.min_xyz, p <= cloud.max_xyz).all(dim=1))


class RandomCubicCrop(Augmentation):
    """augmentations.py:75-89: the cube of edge `size` around a random point (the training configuration's augmentation)."""

    def __init__(self, size):
        self.size = size

    def __call__(self, cloud: Cloud) -> Cloud:
        centre = cloud.xyz[torch.randint(0, cloud.xyz.shape[0], (1,))]
        lo, hi = centre - self.size / 2, centre + self.size / 2
        return cloud.filter(torch.logical_and(cloud.xyz >= lo, cloud.xyz <= hi).all(dim=1))


class RandomDropout(Augmentation):
    """augmentations.py:92-105: a random multiset of the points (sampling WITH replacement, as the reference does)."""

    def __init__(self, max_drop_out):
        self.max_drop_out = max_drop_out

    def __call__(self, cloud: Cloud) -> Cloud:
        dev = cloud.xyz.device
        n = int((1.0 - (self.max_drop_out * torch.rand(1, device=dev))) * cloud.xyz.shape[0])
        return cloud.filter(torch.randint(high=cloud.xyz.shape[0], size=(n, 1), device=dev).squeeze(1))


class AugmentationPipeline(Augmentation):
    def __init__(self, augmentations: Sequence[Augmentation]):
        self.augmentations = list(augmentations)

    def __call__(self, cloud: Cloud) -> Cloud:
        for aug in self.augmentations:
            cloud = aug(cloud)
        return cloud
