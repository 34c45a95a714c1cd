"""Point-cloud container with the reference's field names and method surface.

Mirrors `smart_tree/data_types/cloud.py:19-264` (fields :21-28; `filter` :72-95;
`filter_by_class` :97-103; `to_device` :138-161; `translate` :197-198; `root_idx` :204-206;
`bbox` :222-227; `medial_pts` :229-231; `from_numpy` :233-252; `radius` :254-256).  The open3d
visualisation helpers are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from pathlib import Path
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

_PER_POINT = ("xyz", "rgb", "medial_vector", "branch_direction", "branch_ids", "class_l")

_CLASS_LISTS: dict = {}


def _class_tensor(classes, device) -> torch.Tensor:
    """`classes` as a device tensor, made once per (list, device): `torch.as_tensor(list, device=gpu)` is a pageable host-to-device
    copy the HOST waits for -- behind everything already enqueued on the stream, i.e. in `filter_by_class` behind the whole network --
    so the kernels of the next stage were only enqueued once the GPU had run dry (a ~35 us hole per call in the kernel trace)."""
    if torch.is_tensor(classes):
        return classes.to(device)
    key = (tuple(int(c) for c in classes), str(device))
    t = _CLASS_LISTS.get(key)
    if t is None:
        t = _CLASS_LISTS[key] = torch.as_tensor(list(key[0]), device=device)
    return t


@dataclass
class Cloud:
    xyz: torch.Tensor
    rgb: Optional[torch.Tensor] = None
    medial_vector: Optional[torch.Tensor] = None
    branch_direction: Optional[torch.Tensor] = None
    branch_ids: Optional[torch.Tensor] = None
    class_l: Optional[torch.Tensor] = None
    filename: Optional[Path] = None
    # Additive: a BATCH of independent clouds in one set of arrays (Cloud.collate): cloud b owns the points
    # [seg_off[b], seg_off[b+1]) -- the batch index the reference writes into coords[:,0] (model/sparse.py:40-61), carried
    # from the input clouds to the skeletons so that a whole batch goes through ONE set of kernel launches.
    seg_off: Optional[torch.Tensor] = None  # [B+1] int32 on the cloud's device; None = a single cloud

    def __post_init__(self):
        n = self.xyz.shape[0]
        if self.xyz.ndim != 2 or self.xyz.shape[1] != 3:
            raise TypeError(f"xyz must be [N,3], got {tuple(self.xyz.shape)}")
        for name, width in (("rgb", 3), ("medial_vector", 3), ("branch_direction", 3), ("branch_ids", 1), ("class_l", 1)):
            t = getattr(self, name)
            if t is not None and tuple(t.shape) != (n, width):
                raise TypeError(f"{name} must be [{n},{width}], got {tuple(t.shape)}")

    def __len__(self) -> int:
        return self.xyz.shape[0]

    def __str__(self) -> str:
        return (f"Cloud with {len(self)} points, min {self.min_xyz.tolist()}, "
                f"max {self.max_xyz.tolist()}, device {self.xyz.device}")

    # -- generic per-point map -------------------------------------------------------------
    def _map(self, fn) -> "Cloud":
        changed = {name: fn(getattr(self, name)) for name in _PER_POINT if getattr(self, name) is not None}
        return replace(self, **changed)

    def filter(self, mask, assume_sorted: bool = False) -> "Cloud":
        """Boolean-mask or index gather of every per-point field (reference cloud.py:72-95).  A batch of clouds (`seg_off`)
        can only be cut by a selection that keeps the points in order -- a boolean mask, or a strictly increasing index
        (checked, unless the caller vouches for it with `assume_sorted`): anything else would mix the clouds."""
        mask = mask.to(self.xyz.device)
        if mask.dtype == torch.bool:  # one compaction (one host sync) for all fields instead of one per field
            mask = mask.nonzero().view(-1)
        elif self.seg_off is not None and not assume_sorted and mask.numel() > 1 and not bool((mask[1:] > mask[:-1]).all()):
            raise ValueError("Cloud.filter: a batch of clouds (seg_off) needs a boolean mask or a strictly increasing index; "
                             "split() the batch before reordering / resampling its points")
        out = self._map(lambda t: t.index_select(0, mask))
        if self.seg_off is not None:  # batched: the clouds' new ranges (`mask` must keep the points in order)
            out.seg_off = torch.searchsorted(mask, self.seg_off.to(mask.dtype)).to(torch.int32)
        return out

    # -- batches of independent clouds -------------------------------------------------------
    @property
    def n_seg(self) -> int:
        return 1 if self.seg_off is None else int(self.seg_off.shape[0]) - 1

    @staticmethod
    def collate(clouds) -> "Cloud":
        """B clouds -> one Cloud whose arrays are the concatenation, with `seg_off` marking the clouds."""
        clouds = list(clouds)
        dev = clouds[0].xyz.device
        sizes = [0] + [len(c) for c in clouds]
        kw = {}
        for name in _PER_POINT:
            parts = [getattr(c, name) for c in clouds]
            if all(p is not None for p in parts):
                kw[name] = torch.cat(parts)
        off = torch.tensor(sizes, dtype=torch.int64).cumsum(0).to(torch.int32)
        return Cloud(seg_off=off.to(dev, non_blocking=True), **kw)

    def split(self):
        """The clouds of a batch as separate Cloud objects (views; one host read of the offsets)."""
        if self.seg_off is None:
            return [self]
        off = self.seg_off.tolist()
        out = []
        for a, b in zip(off[:-1], off[1:]):
            out.append(Cloud(**{name: getattr(self, name)[a:b] for name in _PER_POINT if getattr(self, name) is not None}))
        return out

    def filter_by_class(self, classes) -> "Cloud":
        wanted = _class_tensor(classes, self.class_l.device)
        return self.filter(torch.isin(self.class_l, wanted).view(-1))

    def to_device(self, device) -> "Cloud":
        out = self._map(lambda t: t.to(device))
        if self.seg_off is not None:
            out.seg_off = self.seg_off.to(device)
        return out

    def cpu(self) -> "Cloud":
        return self.to_device(torch.device("cpu"))

    def pin_memory(self) -> "Cloud":
        return self._map(lambda t: t.pin_memory())

    def cat(self) -> torch.Tensor:
        return torch.cat((self.xyz, self.rgb), 1)

    def voxel_down_sample(self, voxel_size) -> "Cloud":
        """cloud.py:190-192 with util/misc.py:61-79: one point per occupied voxel of the lattice floor(xyz / voxel_size), voxels in
        lexicographic order, the first point (input order) of each -- except that the reference's index arithmetic skips the
        first voxel (`ind_sorted[cum_sum[1:]]`); kept."""
        q = torch.div(self.xyz, voxel_size, rounding_mode="floor")
        _, inverse, counts = torch.unique(q, dim=0, sorted=True, return_inverse=True, return_counts=True)
        _, by_voxel = torch.sort(inverse, stable=True)
        starts = counts.cumsum(0)[:-1]
        return self.filter(by_voxel[starts])

    # -- geometry (these drop every field but xyz/rgb, as the reference does: cloud.py:194-202)
    #    -- the batch offsets are kept: these ops leave the point order alone)
    def scale(self, factor) -> "Cloud":
        return Cloud(self.xyz * factor, self.rgb, seg_off=self.seg_off)

    def translate(self, offset) -> "Cloud":
        return Cloud(self.xyz + offset.to(self.xyz.device), self.rgb, seg_off=self.seg_off)

    def rotate(self, rot_mat) -> "Cloud":
        rot_mat = rot_mat.to(device=self.xyz.device, dtype=self.xyz.dtype)
        return Cloud(self.xyz @ rot_mat, self.rgb, seg_off=self.seg_off)

    @property
    def max_xyz(self) -> torch.Tensor:
        return self.xyz.max(0)[0]

    @property
    def min_xyz(self) -> torch.Tensor:
        return self.xyz.min(0)[0]

    @property
    def bbox(self):
        """(centre, half-extent), reference cloud.py:222-227."""
        half = (self.max_xyz - self.min_xyz) / 2
        return self.min_xyz + half, half

    @property
    def root_idx(self) -> int:
        """Index of the lowest point (first minimum of y), reference cloud.py:204-206."""
        return int(torch.argmin(self.xyz[:, 1]).item())

    @property
    def number_classes(self) -> int:
        return 1 if self.class_l is None else int(self.class_l.max().item()) + 1

    @property
    def medial_pts(self) -> torch.Tensor:
        return self.xyz + self.medial_vector

    @property
    def radius(self) -> torch.Tensor:
        """|medial_vector| (reference cloud.py:254-256 writes `.pow(2).sum(1).sqrt()`, whose summation
        order is backend dependent; the explicit (x*x + y*y) + z*z below is the same value up to the
        last bit and is identical on every device and in the oracle)."""
        v = self.medial_vector
        return ((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]).sqrt()

    @property
    def direction(self) -> torch.Tensor:
        return F.normalize(self.medial_vector)

    @staticmethod
    def from_numpy(**arrays) -> "Cloud":
        """Keys as in the reference's .npz clouds (cloud.py:233-252); legacy key `vector`."""
        known = {f.name for f in fields(Cloud)} - {"filename"}
        kw = {}
        for key, value in arrays.items():
            if key in known:
                kw[key] = torch.as_tensor(np.asarray(value)).float()
            elif key == "vector":
                kw["medial_vector"] = torch.as_tensor(np.asarray(value))
        return Cloud(**kw)


def _same_device(a: torch.device, b) -> bool:
    b = torch.device(b)
    return a.type == b.type and (b.index is None or a.index is None or a.index == b.index)


class MaskedCloud(Cloud):
    """`base.filter(mask)` (boolean mask) not carried out yet -- what `ModelInference.forward` returns.  Every field reads
    as the filtered cloud's (the compaction happens on first access), but `filter_by_class`, the next thing the pipeline
    does (pipeline.py:71), folds the class test into the pending mask: one compaction and one host round trip for the
    count instead of two (a blocking read-back costs ~1 ms beside other clouds' kernels).  The
    result is the same cloud: both filters keep the surviving points in input order."""

    def __init__(self, base: Cloud, mask: torch.Tensor):
        assert mask.dtype == torch.bool and mask.shape[0] == len(base)
        self.__dict__.update(_base=base, _mask=mask.to(base.xyz.device), _real=None)

    def _cloud(self) -> Cloud:
        if self.__dict__["_real"] is None:
            self.__dict__["_real"] = self.__dict__["_base"].filter(self.__dict__["_mask"])
        return self.__dict__["_real"]

    def filter_by_class(self, classes) -> Cloud:
        """Still pending (round 5): the class test joins the mask, and `Skeletonizer.forward`, the next thing the pipeline does,
        folds its outlier filter into the same selection (`pending()`): ONE compaction and one host round trip for all three."""
        base, mask = self.__dict__["_base"], self.__dict__["_mask"]
        if self.__dict__["_real"] is not None or base.class_l is None:
            return self._cloud().filter_by_class(classes)
        wanted = _class_tensor(classes, base.class_l.device)
        return MaskedCloud(base, mask & torch.isin(base.class_l, wanted).view(-1))

    def pending(self):
        """(base cloud, boolean mask) while the selection has not been carried out, else None."""
        return None if self.__dict__["_real"] is not None else (self.__dict__["_base"], self.__dict__["_mask"])

    def _map(self, fn) -> Cloud:
        return self._cloud()._map(fn)

    def to_device(self, device) -> Cloud:
        if self.__dict__["_real"] is None and _same_device(self.__dict__["_base"].xyz.device, device):
            return self
        return self._cloud().to_device(device)

    def __repr__(self):
        return f"MaskedCloud({self._cloud()!r})"


for _name in [f.name for f in fields(Cloud)]:
    setattr(MaskedCloud, _name, property(lambda self, _name=_name: getattr(self._cloud(), _name)))
del _name

