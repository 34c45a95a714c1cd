"""ModelInference: cloud -> per-voxel medial vector + class (reference smart_tree/model/model_inference.py).

Same constructor keywords and `forward(cloud, return_masked=True) -> Cloud` contract (:22-100).
Differences in HOW: the weights are read as a state_dict only (the pickled module `model_path`
points at is never unpickled -- SURVEY.md 5.4), blocking + voxelisation run on the GPU in one call,
every block goes through the network in one batch, and nothing visits the host between the input
cloud and the output Cloud (the reference does a `.cpu()` per batch, :73-78).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from .. import profiling
from ..data_types.cloud import Cloud, MaskedCloud
from ..dataset.dataset import SingleTreeInference, voxelize_cloud
from .model import Smart_Tree
from .sparse import sparse_from_batch

_WEIGHTS_DIR = Path(__file__).resolve().parent / "weights"


def _load_state_dict(weights_path):
    """Accepts the reference's `*_model_weights.pt` (torch state_dict) or this package's `.npz` fixture;
    a reference path that does not exist here falls back to the bundled fixture of the same run name."""
    p = Path(str(weights_path))
    if not p.exists():
        bundled = _WEIGHTS_DIR / (p.name.replace("_model_weights.pt", "").replace("_model.pt", "").replace(".npz", "") + ".npz")
        if not bundled.exists():
            raise FileNotFoundError(f"weights not found: {weights_path}")
        p = bundled
    if p.suffix == ".npz":
        with np.load(p) as z:
            return {k: torch.from_numpy(z[k]) for k in z.files}
    return torch.load(str(p), map_location="cpu", weights_only=True)


def load_model(model_path, weights_path, device=torch.device("cuda:0"), fp16: bool = False):
    """Reference signature (model_inference.py:11-16).  `model_path` (full pickled nn.Module) is ignored:
    the graph is rebuilt from the state_dict's keys.  fp16: see Smart_Tree (additive keyword)."""
    return Smart_Tree(_load_state_dict(weights_path), device=device, fp16=fp16).eval()


class ModelInference:
    def __init__(self, model_path, weights_path, voxel_size: float, block_size: float, buffer_size: float,
                 num_workers=8, batch_size=4, device=torch.device("cuda:0"), verbose=False, fp16: bool = False,
                 blocking: str = "blocks"):
        """blocking (additive keyword): "blocks" = the reference's scheme (4 m cubes + halo, every block on its
        own voxel grid, halo voxels evaluated and thrown away: ~1.5x the voxels); "whole" = OPT-IN approximate mode: the cloud
        is voxelised once on one grid anchored at its own bounding box (st_voxelize_cloud_seg) and the network sees every
        voxel exactly once with its true neighbourhood.  NOT result-preserving: the reference's grids are anchored per block
        (dataset.py:196-212) and its halo (0.4 m) is shorter than the network's receptive field, so the per-block outputs
        are themselves an approximation of this mode's (tests/test_whole_cloud_mode.py quantifies the difference)."""
        if blocking not in ("blocks", "whole"):
            raise ValueError(f"ModelInference: blocking must be 'blocks' or 'whole', got {blocking!r}")
        self.blocking = blocking
        self.device = torch.device(device)
        self.verbose = verbose
        self.voxel_size = voxel_size
        self.block_size = block_size
        self.buffer_size = buffer_size
        self.num_workers = num_workers
        self.batch_size = batch_size
        self.model = load_model(model_path, weights_path, self.device, fp16=fp16)
        # optional callable, invoked at the end of every forward() when the network's last kernel has been enqueued (see
        # Skeletonizer.on_wide_phase_done: a caller with several batches in flight can schedule their phases with the two hooks)
        self.on_network_done = None
        # [M] int64, set by every forward(): the input point each voxel row stands for (index into the cloud / collated batch it
        # was given) -- lets a caller join any other per-point field to the labelled cloud (bench.py joins the generator's
        # ground-truth medial vectors for its representative end-to-end figure)
        self.last_point_index = None
        if self.verbose:
            print("Model Loaded Succesfully")

    def forward(self, cloud: Cloud, return_masked: bool = True) -> Cloud:
        cloud = cloud.to_device(self.device)
        if cloud.rgb is None:
            cloud = Cloud(cloud.xyz, torch.zeros_like(cloud.xyz), seg_off=cloud.seg_off)
        with profiling.stage("voxelize"):
            if self.blocking == "whole":  # SURVEY 8f.2, opt-in: one grid per cloud, no halo duplicates, every voxel is "inner"
                vb = voxelize_cloud(cloud.xyz, cloud.rgb, self.voxel_size, seg_off=cloud.seg_off)
                if vb.n_seg > 1:  # batch index = cloud: per-cloud spatial extents in the strided rulebooks
                    vb.blk_seg = torch.arange(vb.n_seg, dtype=torch.int32, device=self.device)
            else:
                vb = SingleTreeInference(cloud, self.voxel_size, self.block_size, self.buffer_size).batch
        # every block of the cloud -- of every cloud of a batch (Cloud.collate) -- in ONE collated batch
        # blocks mode: batch indices < number of blocks, voxel coordinates < round(block + 2 halos) / voxel (csrc/voxelize.hip
        # k_vx_block_grid) -- the bounds the network's brick rulebooks are sized from (verified on the device)
        hint = None
        if self.blocking == "blocks" and vb.block_centres.shape[0] > 0:
            hint = (int(vb.block_centres.shape[0]), int(round((self.block_size + 2 * self.buffer_size) / self.voxel_size)) + 2)
        sparse_input = sparse_from_batch(vb.feats[:, :3].contiguous(), vb.coords, device=self.device, blk_seg=vb.blk_seg,
                                         n_seg=vb.n_seg, brick_hint=hint)
        # radius / direction / class_l come out exactly as model.forward(sparse_input) gives them;
        # exp(radius)*direction and argmax (reference :87-88) are fused into the head kernel
        with profiling.stage("unet"):
            _, _, _, mv, cls = self.model.forward_fused_tail(sparse_input)
        masks = vb.mask
        self.last_point_index = vb.point_index
        lc = Cloud(xyz=sparse_input.features, rgb=vb.feats[:, 3:6].contiguous(), medial_vector=mv, class_l=cls,
                   seg_off=vb.seg_vox_off)
        if self.on_network_done is not None:
            self.on_network_done()
        # the inner-block filter (reference :97-100) is handed on as a pending mask: Pipeline's filter_by_class folds into it
        return MaskedCloud(lc, masks) if return_masked else lc

    @staticmethod
    def from_cfg(cfg):
        return ModelInference(model_path=cfg.model_path, weights_path=cfg.weights_path, voxel_size=cfg.voxel_size,
                              block_size=cfg.block_size, buffer_size=cfg.buffer_size, num_workers=cfg.num_workers,
                              batch_size=cfg.batch_size, blocking=getattr(cfg, "blocking", "blocks"))
