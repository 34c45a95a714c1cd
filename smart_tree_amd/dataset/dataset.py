"""Inference-side blocking + voxelisation, device resident.

Reference: `SingleTreeInference` (smart_tree/dataset/dataset.py:144-229) cuts the cloud into 4 m
blocks with a 0.4 m halo through a host loop of full-cloud masks and per-block D->H copies, then
voxelises each block on the CPU with spconv's `PointToVoxel` inside a DataLoader
(`load_dataloader`, :232-242) and `batch_collate`s <=4 blocks (smart_tree/model/sparse.py:40-61).
Here the whole of it is one call into `st_voxelize_blocks` (csrc/voxelize.hip); every block of
the cloud lands in a single batch (eval-mode BatchNorm: batching cannot change a value) in the
deterministic order "blocks as torch.unique sorts them, voxels by first point" -- the order the
reference would produce with its accidental `shuffle=True` removed.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib
from ..data_types.cloud import Cloud


_VOXELS_PER_POINT = {}  # (voxel, block, buffer) -> voxels per input point of the last call: sizes the next call's hash table


@dataclass
class VoxelBatch:
    feats: torch.Tensor  # [M,6] float32 representative point (xyz, rgb)
    coords: torch.Tensor  # [M,4] int32 (block, z, y, x)
    mask: torch.Tensor  # [M] bool: representative point inside the un-buffered block
    point_index: torch.Tensor  # [M] int64 index of the representative point in the input cloud
    block_centres: torch.Tensor  # [B,3] float32
    # batched call (xyz holds several clouds, Cloud.collate): cloud of every block, first voxel / block of every cloud
    blk_seg: Optional[torch.Tensor] = None  # [B] int32
    seg_vox_off: Optional[torch.Tensor] = None  # [n_seg+1] int32
    seg_blk_off: Optional[torch.Tensor] = None  # [n_seg+1] int32
    n_seg: int = 1


def voxelize_blocks(xyz: torch.Tensor, rgb: Optional[torch.Tensor], voxel_size: float, block_size: float = 4,
                    buffer_size: float = 0.4, min_points: int = 20, max_blocks: int = 4096,
                    seg_off: Optional[torch.Tensor] = None) -> VoxelBatch:
    """seg_off ([B+1] int32, device): xyz holds B independent clouds; blocks are numbered cloud by cloud and the part of
    every cloud equals the one-cloud result (block index shifted, `point_index` into the batched array)."""
    L = _lib.lib()
    dev = xyz.device
    xyz = xyz.contiguous().float()
    rgb = rgb.contiguous().float() if rgb is not None else None
    n = xyz.shape[0]
    nseg = 1 if seg_off is None else int(seg_off.shape[0]) - 1
    return _voxelize_blocks(xyz, rgb, n, nseg, seg_off, voxel_size, block_size, buffer_size, min_points, max_blocks)


def _voxelize_blocks(xyz, rgb, n, nseg, seg_off, voxel_size, block_size, buffer_size, min_points, max_blocks):
    L = _lib.lib()
    dev = xyz.device
    per_cloud_blocks = max_blocks
    max_blocks = min(max_blocks * nseg, 65535)
    n_vox, n_blk = ctypes.c_int64(0), ctypes.c_int64(0)
    i32 = lambda k: torch.empty((k,), dtype=torch.int32, device=dev)
    blk_seg, seg_vox, seg_blk = (i32(max_blocks), i32(nseg + 1), i32(nseg + 1)) if nseg > 1 else (None, None, None)
    # Output capacity = hash-table size / 2: a table sized for the worst case (a point sits in <= 8 halo cubes) is ~40x
    # larger than what a tree cloud needs, and every probe into it is a cache miss.  First guess: 1.5 x the voxels-per-point
    # ratio the last call with these parameters saw (0.5 the first time); the kernel flags an overflow, then the worst-case
    # sizes follow.
    key = (float(voxel_size), float(block_size), float(buffer_size))
    worst = (8 if 2 * buffer_size < block_size else 27) * n + 1024  # halo cubes a point can sit in
    guess = int(n * 1.5 * _VOXELS_PER_POINT.get(key, 1.0 / 3.0)) + 4096
    for cap in sorted({min(guess, worst), min(3 * n + 1024, worst), worst}):
        feats = torch.empty((cap, 6), dtype=torch.float32, device=dev)
        coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        mask = torch.empty((cap,), dtype=torch.uint8, device=dev)
        pidx = torch.empty((cap,), dtype=torch.int64, device=dev)
        centres = torch.empty((max_blocks, 3), dtype=torch.float32, device=dev)
        ws = _lib.workspace(L.st_voxelize_workspace_bytes_seg(n, max_blocks, cap, nseg), dev)
        rc = L.st_voxelize_blocks_seg(_lib.ptr(xyz), _lib.ptr(rgb), n, _lib.ptr(seg_off), nseg, float(voxel_size),
                                      float(block_size), float(buffer_size), int(min_points), int(max_blocks), cap,
                                      _lib.ptr(feats), _lib.ptr(coords), _lib.ptr(mask), _lib.ptr(pidx), _lib.ptr(centres),
                                      _lib.ptr(blk_seg), _lib.ptr(seg_vox), _lib.ptr(seg_blk), ctypes.byref(n_vox),
                                      ctypes.byref(n_blk), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
        if rc == 0 or b"exceed max_voxels" not in L.st_last_error():
            break
    if rc != 0 and (b"bounding box of a cloud has more than" in L.st_last_error() or b"exceed max_blocks" in L.st_last_error()) \
            and per_cloud_blocks * nseg < 65535:
        # a plot larger than the default block table (8 x max_blocks cells, >= 32768: e.g. a stray point far away, or hundreds
        # of metres of forest): the reference's torch.unique handles any extent -- retry with a larger table
        return _voxelize_blocks(xyz, rgb, n, nseg, seg_off, voxel_size, block_size, buffer_size, min_points,
                                min(per_cloud_blocks * 8, 65535))
    _lib.check(rc)
    m, b = n_vox.value, n_blk.value
    if n > 0:
        _VOXELS_PER_POINT[key] = max(m / n, 1e-3)
    return VoxelBatch(feats[:m], coords[:m], mask[:m].bool(), pidx[:m], centres[:b],
                      blk_seg[:b] if blk_seg is not None else None, seg_vox, seg_blk, nseg)


def voxelize_cloud(xyz: torch.Tensor, rgb: Optional[torch.Tensor], voxel_size: float,
                   seg_off: Optional[torch.Tensor] = None) -> VoxelBatch:
    """Whole-cloud voxelisation (st_voxelize_cloud_seg): every cloud is one block spanning its own bounding box, one
    representative point per voxel -- TreeDataset.process_cloud's PointToVoxel call (dataset.py:103-131) and, for a batch
    (seg_off), batch_collate's batch column (model/sparse.py:40-61).  `point_index` gathers any other per-point feature."""
    L = _lib.lib()
    dev = xyz.device
    xyz = xyz.contiguous().float()
    rgb = rgb.contiguous().float() if rgb is not None else None
    n = xyz.shape[0]
    nseg = 1 if seg_off is None else int(seg_off.shape[0]) - 1
    cap = n + 8  # a voxel needs a point
    feats = torch.empty((cap, 6), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    mask = torch.empty((cap,), dtype=torch.uint8, device=dev)
    pidx = torch.empty((cap,), dtype=torch.int64, device=dev)
    seg_vox = torch.empty((nseg + 1,), dtype=torch.int32, device=dev) if nseg > 1 else None
    n_vox = ctypes.c_int64(0)
    ws = _lib.workspace(L.st_voxelize_cloud_workspace_bytes(n, cap, nseg), dev)
    _lib.check(L.st_voxelize_cloud_seg(_lib.ptr(xyz), _lib.ptr(rgb), n, _lib.ptr(seg_off), nseg, float(voxel_size), cap,
                                       _lib.ptr(feats), _lib.ptr(coords), _lib.ptr(mask), _lib.ptr(pidx), _lib.ptr(seg_vox),
                                       ctypes.byref(n_vox), _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    m = n_vox.value
    return VoxelBatch(feats[:m], coords[:m], mask[:m].bool(), pidx[:m], torch.zeros((nseg, 3), dtype=torch.float32, device=dev),
                      None, seg_vox, None, nseg)


def at_least_2d(t: torch.Tensor) -> torch.Tensor:
    """util/misc.py:13-21."""
    return t.unsqueeze(1) if t.dim() == 1 else t


class TreeDataset:
    """Training / evaluation-side dataset (reference dataset.py:18-141): labelled clouds listed in a json split file, one
    augmentation pipeline, whole-cloud voxelisation with one representative point per voxel whose input AND target features
    are carried along.  Same constructor and item layout as the reference: `((input_feats, target_feats), coords, loss_mask,
    filename)` with coords [M,4] = (0, z, y, x); `model.sparse.batch_collate` writes the sample index into column 0.
    The voxeliser is st_voxelize_cloud_seg (csrc/voxelize.hip) instead of spconv's PointToVoxel; the feature columns are one
    gather by the representative point's index."""

    def __init__(self, voxel_size, json_path, directory, mode: str, input_features, target_features, augmentation=None,
                 cache: bool = False, device=None):
        import json
        from pathlib import Path

        self.voxel_size = voxel_size
        self.mode = mode
        self.augmentation = augmentation
        self.directory = directory
        self.device = device if device is not None else torch.device("cuda:0")
        self.input_features = list(input_features)
        self.target_features = list(target_features)
        if not Path(json_path).is_file():
            raise AssertionError(f"json metadata does not exist at '{json_path}'")
        if mode not in ("train", "validation", "test"):
            raise ValueError(f"TreeDataset: mode must be train / validation / test, got {mode!r}")
        with open(json_path) as f:
            self.tree_paths = json.load(f)[mode]
        missing = [p for p in self.tree_paths if not Path(f"{self.directory}/{p}").is_file()]
        if missing:
            raise AssertionError(f"Missing {len(missing)} files: {missing}")
        self.cache = {} if cache else None

    def load(self, filename) -> Cloud:
        from ..util.file import load_cloud

        if self.cache is None:
            return load_cloud(filename)
        if filename not in self.cache:
            cld = load_cloud(filename)
            self.cache[filename] = cld.pin_memory() if torch.cuda.is_available() else cld
        return self.cache[filename]

    def __getitem__(self, idx):
        from pathlib import Path

        return self.process_cloud(self.load(Path(f"{self.directory}/{self.tree_paths[idx]}")), self.tree_paths[idx])

    def process_cloud(self, cld: Cloud, filename):
        cld = cld.to_device(self.device)
        if self.augmentation is not None:
            cld = self.augmentation(cld)
        inputs = torch.cat([at_least_2d(getattr(cld, a)) for a in self.input_features], dim=1)
        targets = torch.cat([at_least_2d(getattr(cld, a)) for a in self.target_features], dim=1)
        if inputs.shape[0] == 0:
            raise AssertionError(f"Empty cloud after augmentation: {filename}")
        vb = voxelize_cloud(cld.xyz, None, self.voxel_size)
        rep = vb.point_index
        loss_mask = torch.ones(rep.shape[0], dtype=torch.bool, device=rep.device)
        return (inputs.index_select(0, rep), targets.float().index_select(0, rep)), vb.coords, loss_mask, filename

    def __len__(self) -> int:
        return len(self.tree_paths)


class SingleTreeInference:
    """Same constructor surface as the reference class (dataset.py:145-164)."""

    def __init__(self, cloud: Cloud, voxel_size: float, block_size: float = 4, buffer_size: float = 0.4,
                 min_points: int = 20, file_name=None, device=None):
        self.cloud = cloud
        self.voxel_size = voxel_size
        self.block_size = block_size
        self.buffer_size = buffer_size
        self.min_points = min_points
        self.file_name = file_name
        self.batch = voxelize_blocks(cloud.xyz, cloud.rgb, voxel_size, block_size, buffer_size, min_points,
                                     seg_off=cloud.seg_off)
        self.block_centres = self.batch.block_centres

    def __len__(self) -> int:
        return self.block_centres.shape[0]

    def __getitem__(self, idx):
        """(feats, coords, mask, filename) of block idx, coords[:,0] = 0 as dataset.py:218-226."""
        sel = self.batch.coords[:, 0] == idx
        coords = self.batch.coords[sel].clone()
        coords[:, 0] = 0
        return self.batch.feats[sel], coords, self.batch.mask[sel], self.file_name


def load_dataloader(cloud: Cloud, voxel_size: float, block_size: float, buffer_size: float, num_workers: float,
                    batch_size: float):
    """Reference signature (dataset.py:232-242).  Yields ONE collated batch holding every block;
    `num_workers` / `batch_size` are accepted for drop-in compatibility and ignored."""
    ds = SingleTreeInference(cloud, voxel_size, block_size, buffer_size)
    b = ds.batch
    return [(b.feats, b.coords, b.mask, ds.file_name)]
