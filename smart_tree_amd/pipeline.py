"""Pipeline: point cloud -> skeleton (reference smart_tree/pipeline.py:13-106).

Constructor keywords, defaults and `process_cloud(path=None, cloud=None)` / `post_process(skeleton)`
match the reference.  Additive change: `process_cloud` returns the `DisjointTreeSkeleton` (the
reference only views / saves it).  Viewing needs open3d and is out of scope; `save_outputs` writes
dependency-free files (util/file.py).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import profiling
from .data_types.cloud import Cloud
from .data_types.tree import DisjointTreeSkeleton
from .util.mesh import skeleton_mesh, write_ply_mesh
from .util.file import load_cloud, save_skeleton, save_skeleton_npz, write_ply_points, write_ply_skeleton


class Pipeline:
    def __init__(self, preprocessing, model_inference, skeletonizer, repair_skeletons=False, smooth_skeletons=False,
                 smooth_kernel_size=0, prune_skeletons=False, min_skeleton_radius=0.0, min_skeleton_length=1000,
                 view_model_output=False, view_skeletons=False, save_outputs=False, save_path="/", branch_classes=[0],
                 cmap=[[1, 0, 0], [0, 1, 0]], device=torch.device("cuda:0")):
        self.preprocessing = preprocessing
        self.model_inference = model_inference
        self.skeletonizer = skeletonizer
        self.repair_skeletons = repair_skeletons
        self.smooth_skeletons = smooth_skeletons
        self.smooth_kernel_size = smooth_kernel_size
        self.prune_skeletons = prune_skeletons
        self.min_skeleton_radius = min_skeleton_radius
        self.min_skeleton_length = min_skeleton_length
        self.view_model_output = view_model_output
        self.view_skeletons = view_skeletons
        self.save_outputs = save_outputs
        self.save_path = save_path
        self.branch_classes = branch_classes
        self.cmap = np.asarray(cmap)
        self.device = torch.device(device)
        self.last_labelled_cloud = None

    def process_cloud(self, path: Path = None, cloud: Cloud = None) -> DisjointTreeSkeleton:
        cloud = load_cloud(path) if path is not None else cloud
        cloud = cloud.to_device(self.device)
        with profiling.stage("preprocess"):
            cloud = self.preprocessing(cloud)
        lc = self.model_inference.forward(cloud).to_device(self.device)
        self.last_labelled_cloud = lc
        with profiling.stage("class_filter"):
            branch_cloud = lc.filter_by_class(self.branch_classes)
        skeleton = self.skeletonizer.forward(branch_cloud)
        with profiling.stage("post_process"):
            self.post_process(skeleton)
            skeleton.skeletons  # materialise: device post-processing kernel, one D->H copy, BranchSkeleton objects
        if self.view_model_output or self.view_skeletons:
            raise NotImplementedError("viewing needs open3d, which is out of scope of smart_tree_amd")
        if self.save_outputs:  # reference pipeline.py:85-93 (skeleton.ply / cloud.ply; meshes need open3d)
            sp = Path(self.save_path)
            save_skeleton_npz(sp / "skeleton.npz", skeleton)
            for tree in skeleton.skeletons:  # the reference's own per-tree layout (util/file.py:73-94), readable by its load_skeleton
                if tree.branches:
                    save_skeleton(tree, sp / f"skeleton_{tree._id}.npz")
            write_ply_skeleton(sp / "skeleton.ply", skeleton)
            verts, tris = skeleton_mesh(skeleton)  # reference pipeline.py:90: mesh.ply = the skeleton's tube mesh
            write_ply_mesh(sp / "mesh.ply", verts, tris)
            write_ply_points(sp / "cloud.ply", lc.xyz.cpu().numpy(), lc.rgb.cpu().numpy() if lc.rgb is not None else None)
        return skeleton

    def process_clouds(self, clouds) -> list:
        """B independent clouds through ONE set of kernel launches (additive; the reference's unit of work is a batch
        as well: model/sparse.py:40-61 writes the batch index into coords[:,0], model_inference.py:62-78 runs a batch per
        forward -- here the batch index is carried through every stage: blocks / voxels, network, kNN graph, components,
        SSSP, branch selection, post-processing).  Returns one DisjointTreeSkeleton per cloud, each identical to
        `process_cloud` of that cloud alone (tests/test_batch.py)."""
        clouds = [c.to_device(self.device) for c in clouds]
        if not clouds:
            return []
        batch = Cloud.collate([Cloud(c.xyz, c.rgb if c.rgb is not None else torch.zeros_like(c.xyz)) for c in clouds])
        with profiling.stage("preprocess"):
            batch = self.preprocessing(batch)
        if batch.n_seg != len(clouds):  # an augmentation that rebuilt the Cloud without its batch offsets would merge the inputs
            raise RuntimeError(f"process_clouds: preprocessing returned {batch.n_seg} cloud(s) for a batch of {len(clouds)}")
        lc = self.model_inference.forward(batch).to_device(self.device)
        self.last_labelled_cloud = lc
        with profiling.stage("class_filter"):
            branch_cloud = lc.filter_by_class(self.branch_classes)
        skeleton = self.skeletonizer.forward(branch_cloud)
        with profiling.stage("post_process"):
            self.post_process(skeleton)
            parts = skeleton.split()  # device post-processing of all clouds, ONE device-to-host copy, per-cloud views
        if len(parts) != len(clouds):
            raise RuntimeError(f"process_clouds: {len(parts)} skeleton(s) for a batch of {len(clouds)} clouds")
        if self.view_model_output or self.view_skeletons:
            raise NotImplementedError("viewing needs open3d, which is out of scope of smart_tree_amd")
        return parts

    def post_process(self, skeleton: DisjointTreeSkeleton) -> None:
        if self.prune_skeletons:
            skeleton.prune(min_length=self.min_skeleton_length, min_radius=self.min_skeleton_radius)
        if self.repair_skeletons:
            skeleton.repair()
        if self.smooth_skeletons:
            skeleton.smooth(self.smooth_kernel_size)
