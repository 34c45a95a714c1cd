"""Pipeline.process_cloud on the HIP path vs oracle/pipeline_oracle.py, stage by stage."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import unet_oracle as uo
from smart_tree_amd import cli
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
from smart_tree_amd.model.model_inference import ModelInference
from smart_tree_amd.pipeline import Pipeline
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

ROOT = Path(__file__).resolve().parents[1]
WEIGHTS = ROOT / "smart_tree_amd" / "model" / "weights" / "noble-elevator-58.npz"


def _pipeline(device, voxel=0.03, **kw):
    mi = ModelInference("unused_model.pt", WEIGHTS, voxel_size=voxel, block_size=4, buffer_size=0.4, device=device)
    if device.type == "cpu":  # end-to-end on the sanitizer build: the vector kernels (the matrix-core kernels cost a fiber rendezvous
        mi.model.use_mfma = False  # per instruction there and have their own tests in test_unet.py)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=device)
    sk.block_threads = 128 if device.type == "cpu" else 0
    return Pipeline(AugmentationPipeline([CentreCloud()]), mi, sk, device=device, **kw)


def _same_tree(got, ref):
    assert list(got.branches.keys()) == list(ref.branches.keys())
    for k, rb in ref.branches.items():
        gb = got.branches[k]
        assert gb.parent_id == rb.parent_id
        np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
        np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)


def test_process_cloud_stagewise_parity(backend):
    c = sample_tree_cloud(12000, seed=2, scale=0.5, max_depth=4)
    post = dict(prune_skeletons=True, min_skeleton_radius=0.01, min_skeleton_length=0.02, repair_skeletons=True,
                smooth_skeletons=True, smooth_kernel_size=11)
    pipe = _pipeline(backend, **post)
    cloud = Cloud(xyz=torch.from_numpy(c["xyz"]), rgb=torch.from_numpy(c["rgb"]))
    skeleton = pipe.process_cloud(cloud=cloud)
    lc = pipe.last_labelled_cloud

    # the result gather's fast path (packed host arrays -> table / geometry) against the branch-by-branch walk
    from smart_tree_amd import sharding
    from smart_tree_amd.data_types.tree import DisjointTreeSkeleton
    fast = skeleton.pack(cloud_id=5)
    assert fast is not None
    slow = sharding.pack_skeleton(DisjointTreeSkeleton(list(skeleton.skeletons)), cloud_id=5)  # plain container: the walk
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast[1], slow[1])
    assert skeleton.pack(cloud_id=5) is None  # the objects have been handed out: no shortcut any more

    # stage 1: labelled cloud vs the oracle network (fp32 tolerance; coordinates exact)
    w = uo.load_weights(WEIGHTS)
    ref = po.labelled_cloud(c["xyz"], c["rgb"], w, 0.03, dtype=torch.float64)
    np.testing.assert_array_equal(lc.xyz.cpu().numpy(), ref["xyz"])
    mv = lc.medial_vector.cpu().numpy()
    scale = np.sqrt(np.mean(ref["medial_vector"] ** 2)) + 1e-30
    # north star: 1e-4 relative in float32.  With the shipped checkpoint's BatchNorm statistics (var down to 6.6e-22) two float32
    # evaluation orders differ by more than that, so the bar is 1e-4 or 4x the distance of the float32 ORACLE from the float64
    # one on the same input (the construct of tests/test_full_size.py; plain 1e-4 holds with live weights, tests/test_unet.py)
    ref32 = po.labelled_cloud(c["xyz"], c["rgb"], w, 0.03, dtype=torch.float32)
    base = np.abs(ref32["medial_vector"].astype(np.float64) - ref["medial_vector"]).max() / scale
    err = np.abs(mv - ref["medial_vector"]).max() / scale
    assert err <= max(1e-4, 4 * base), (err, base)
    assert (lc.class_l.cpu().numpy() != ref["class_l"]).mean() < 1e-3

    # stage 2: skeleton + post-processing from the SAME labelled cloud must be identical
    trees = po.skeleton_from_labelled(lc.xyz.cpu().numpy(), mv, lc.class_l.cpu().numpy())
    po.post_process(trees, True, 0.01, 0.02, True, True, 11)
    assert len(skeleton.skeletons) == len(trees)
    for got, rt in zip(skeleton.skeletons, trees):
        _same_tree(got, rt)


def test_yaml_instantiate_mirrors_reference_schema(backend, monkeypatch):
    cfg = cli.load_config([f"pipeline.model_inference.weights_path={WEIGHTS}", "pipeline.model_inference.voxel_size=0.05"])
    for section in ("model_inference", "skeletonizer"):
        cfg["pipeline"][section]["device"] = str(backend)
    cfg["pipeline"]["device"] = str(backend)
    pipe = cli.instantiate(cfg["pipeline"])
    assert isinstance(pipe, Pipeline) and pipe.repair_skeletons and pipe.smooth_kernel_size == 11
    assert pipe.skeletonizer.K == 16 and pipe.model_inference.block_size == 4
