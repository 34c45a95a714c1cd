"""Training / evaluation-side forward path: TreeDataset.process_cloud + batch_collate, the three losses,
evaluate_losses -- HIP path vs oracle/loss_oracle.py and the golden vectors produced by the reference's own loss.py / dataset.py
(tools/make_goldens.py: loss_case, tree_dataset_case)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo
from oracle import unet_oracle as uo
from smart_tree_amd import _lib
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset import augmentations as aug
from smart_tree_amd.dataset.dataset import TreeDataset, voxelize_cloud
from smart_tree_amd.model import loss as L
from smart_tree_amd.model.model import Smart_Tree
from smart_tree_amd.model.sparse import batch_collate
from smart_tree_amd.synthetic import sample_tree_cloud

GOLDEN = Path(__file__).parent / "golden"
WEIGHTS = Path(__file__).resolve().parents[1] / "smart_tree_amd" / "model" / "weights"
CASES = {"plain": dict(mask=False, vector_class=None, target_radius_log=True),
         "masked": dict(mask=True, vector_class=None, target_radius_log=True),
         "masked_vector0": dict(mask=True, vector_class=0, target_radius_log=True),
         "vector1_rawradius": dict(mask=False, vector_class=1, target_radius_log=False)}
LOSS_TOL = 1e-5  # HIP kernel (float32 terms, float64 sums) against the float64 oracle, relative


def _golden_losses():
    g = np.load(GOLDEN / "loss_vectors.npz")
    preds = {k: g[k] for k in ("radius", "direction", "class_l")}
    return g, preds


# ------------------------------------------------------------------------------ oracle pinned by the reference's outputs ---
@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("cls", ["focal", "dice"])
def test_loss_oracle_matches_reference_vectors(case, cls):
    g, preds = _golden_losses()
    kw = CASES[case]
    res = lo.compute_loss(preds, g["targets"], mask=g["mask"] if kw["mask"] else None, target_radius_log=kw["target_radius_log"],
                          vector_class=kw["vector_class"], class_loss=cls)
    ref = g[f"{case}_{cls}"]  # the reference's float32 results
    np.testing.assert_allclose([res["radius"], res["direction"], res["class_l"]], ref, rtol=1e-4)


def _golden_clouds(g):
    out = []
    for k in range(2):
        c = sample_tree_cloud(30_000, seed=int(g[f"seed_{k}"]), scale=0.7, max_depth=4, foliage_fraction=0.3)
        out.append(c)
    return out


def test_process_cloud_oracle_matches_reference_dataset():
    g = np.load(GOLDEN / "tree_dataset.npz")
    parts = []
    for k, c in enumerate(_golden_clouds(g)):
        mv = c["medial_vector"]
        radius = np.sqrt((mv.astype(np.float32) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
        direction = torch.nn.functional.normalize(torch.from_numpy(mv)).numpy()
        targets = np.concatenate([radius[:, None], direction, c["class_l"].reshape(-1, 1)], 1).astype(np.float32)
        i, t, coords, mask, _ = lo.process_cloud(c["xyz"], c["xyz"], targets, 0.03)
        coords[:, 0] = k
        parts.append((i, t, coords, mask))
    np.testing.assert_array_equal(np.concatenate([p[2] for p in parts]), g["coords"])
    np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), g["inputs"])
    np.testing.assert_allclose(np.concatenate([p[1] for p in parts]), g["targets"], rtol=0, atol=1e-6)  # radius: pow/sum/sqrt order
    assert np.concatenate([p[3] for p in parts]).all() and g["mask"].all()


# ---------------------------------------------------------------------------------------------- HIP path vs oracle ---
@pytest.mark.parametrize("case", list(CASES))
def test_fused_losses_match_oracle(backend, case):
    g, preds = _golden_losses()
    kw = CASES[case]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    tp = {k: t(v) for k, v in preds.items()}
    mask = t(g["mask"]) if kw["mask"] else None
    for cls_name, cls_fn in (("focal", L.focal_loss), ("dice", L.dice_loss)):
        ref = lo.compute_loss(preds, g["targets"], mask=g["mask"] if kw["mask"] else None, target_radius_log=kw["target_radius_log"],
                              vector_class=kw["vector_class"], class_loss=cls_name)
        out = L.compute_loss(tp, t(g["targets"]), mask, L.L1Loss, L.cosine_similarity_loss, cls_fn,
                             target_radius_log=kw["target_radius_log"], vector_class=kw["vector_class"])
        for k in ("radius", "direction", "class_l"):
            assert out[k].dtype == torch.float32 and out[k].dim() == 0
            np.testing.assert_allclose(out[k].item(), ref[k], rtol=LOSS_TOL, err_msg=f"{case} {cls_name} {k}")
        # and the reference's float32 numbers themselves
        np.testing.assert_allclose([out[k].item() for k in ("radius", "direction", "class_l")], g[f"{case}_{cls_name}"], rtol=1e-4)


def test_loss_functions_on_their_own_and_foreign_callables(backend):
    """The reference's step-by-step path (boolean selections, then the callables) with this module's functions used one by
    one gives the fused pass's numbers; a foreign callable is simply called."""
    g, preds = _golden_losses()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    tp = {k: t(v) for k, v in preds.items()}
    mask = t(g["mask"])
    fused = L.compute_loss(tp, t(g["targets"]), mask, L.L1Loss, L.cosine_similarity_loss, L.focal_loss, vector_class=0)
    calls = []

    def my_l1(o, tt):
        calls.append(o.shape[0])
        return L.L1Loss(o, tt)

    step = L.compute_loss(tp, t(g["targets"]), mask, my_l1, L.cosine_similarity_loss, L.focal_loss, vector_class=0)
    sel = g["mask"] & (g["targets"][:, 4] == 0)
    assert calls == [int(sel.sum())]
    for k in fused:
        np.testing.assert_allclose(step[k].item(), fused[k].item(), rtol=1e-6)
    assert L.nll_loss(tp["class_l"], t(g["targets"][:, 4])).item() == 0  # loss.py:100-101


def test_loss_edge_cases(backend):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    rng = np.random.RandomState(0)
    n = 300
    preds = {"radius": t(rng.randn(n, 1).astype(np.float32)), "direction": t(rng.randn(n, 3).astype(np.float32)),
             "class_l": t(rng.randn(n, 2).astype(np.float32))}
    targets = np.concatenate([rng.uniform(0.01, 0.1, (n, 1)), rng.randn(n, 3), np.ones((n, 1))], 1).astype(np.float32)
    fns = (L.L1Loss, L.cosine_similarity_loss, L.focal_loss)
    out = L.compute_loss(preds, t(targets), None, *fns, vector_class=0)  # no row of class 0: empty vector selection
    assert np.isnan(out["radius"].item()) and np.isnan(out["direction"].item()) and np.isfinite(out["class_l"].item())
    none = L.compute_loss(preds, t(targets), t(np.zeros(n, bool)), *fns)  # empty loss mask: every mean is over nothing
    assert all(np.isnan(v.item()) for v in none.values())
    bad = targets.copy()
    bad[7, 4] = 2.0  # class id outside the logits: torch's gather raises, so does the kernel
    with pytest.raises(_lib.StError, match="class ids outside"):
        L.compute_loss(preds, t(bad), None, *fns)
    for junk in (np.nan, np.inf, -np.inf):  # a non-finite class id is reported the same way (no undefined float -> int cast)
        bad = targets.copy()
        bad[3, 4] = junk
        with pytest.raises(_lib.StError, match="class ids outside"):
            L.compute_loss(preds, t(bad), None, *fns)


def _write_split(tmp_path, clouds):
    names = []
    for k, c in enumerate(clouds):
        name = f"tree_{k}.npz"
        np.savez(tmp_path / name, xyz=c["xyz"], rgb=c["rgb"], medial_vector=c["medial_vector"], class_l=c["class_l"])
        names.append(name)
    split = tmp_path / "split.json"
    split.write_text(json.dumps({"train": names[:1], "validation": names, "test": names[1:]}))
    return split


def test_tree_dataset_items_and_collate_match_the_reference(backend, tmp_path):
    g = np.load(GOLDEN / "tree_dataset.npz")
    split = _write_split(tmp_path, _golden_clouds(g))
    ds = TreeDataset(0.03, split, tmp_path, "validation", ["xyz"], ["radius", "direction", "class_l"], device=backend)
    assert len(ds) == 2 and len(TreeDataset(0.03, split, tmp_path, "train", ["xyz"], ["radius"], device=backend)) == 1
    items = [ds[0], ds[1]]
    assert items[0][3] == "tree_0.npz" and (items[0][1][:, 0] == 0).all()
    (inputs, targets), coords, mask, names = batch_collate(items)
    np.testing.assert_array_equal(coords.cpu().numpy(), g["coords"])
    np.testing.assert_array_equal(inputs.cpu().numpy(), g["inputs"])
    np.testing.assert_allclose(targets.cpu().numpy(), g["targets"], rtol=0, atol=1e-6)
    assert mask.dtype == torch.bool and mask.all() and names == ("tree_0.npz", "tree_1.npz")
    with pytest.raises(AssertionError, match="Missing 1 files"):
        (tmp_path / "tree_1.npz").unlink()
        TreeDataset(0.03, split, tmp_path, "test", ["xyz"], ["radius"], device=backend)
    with pytest.raises(AssertionError, match="json metadata"):
        TreeDataset(0.03, tmp_path / "nope.json", tmp_path, "test", ["xyz"], ["radius"], device=backend)


def test_voxelize_cloud_batch_equals_one_cloud_at_a_time(backend):
    clouds = [sample_tree_cloud(n, seed=s, scale=0.6, max_depth=3)["xyz"] for n, s in ((7000, 1), (3000, 2), (9000, 3))]
    clouds.insert(2, np.array([[0.5, 0.5, 0.5]], np.float32))  # a one-point cloud: extent 0, grid 0 -> no voxel (PointToVoxel drops it)
    t = lambda a: torch.from_numpy(a).to(backend)
    singles = [voxelize_cloud(t(c), None, 0.04) for c in clouds]
    for c, s in zip(clouds, singles):
        _, _, coords, _, first = lo.process_cloud(c, c, c, 0.04)
        np.testing.assert_array_equal(s.coords.cpu().numpy(), coords)
        np.testing.assert_array_equal(s.point_index.cpu().numpy(), first)
        assert s.mask.all()
    assert singles[2].coords.shape[0] == 0
    batch = Cloud.collate([Cloud(xyz=t(c)) for c in clouds])
    vb = voxelize_cloud(batch.xyz, None, 0.04, seg_off=batch.seg_off)
    off = vb.seg_vox_off.cpu().numpy()
    starts = np.cumsum([0] + [len(c) for c in clouds])
    assert off[0] == 0 and off[-1] == vb.coords.shape[0]
    for k, s in enumerate(singles):
        part = slice(off[k], off[k + 1])
        np.testing.assert_array_equal(vb.coords[part, 1:].cpu().numpy(), s.coords[:, 1:].cpu().numpy())
        assert (vb.coords[part, 0] == k).all()
        np.testing.assert_array_equal(vb.point_index[part].cpu().numpy() - starts[k], s.point_index.cpu().numpy())


def test_augmentations(backend):
    c = sample_tree_cloud(5000, seed=4, scale=0.6, max_depth=3)
    cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(backend), rgb=torch.from_numpy(c["rgb"]).to(backend),
                  medial_vector=torch.from_numpy(c["medial_vector"]).to(backend), class_l=torch.from_numpy(c["class_l"]).to(backend))
    torch.manual_seed(3)
    crop = aug.RandomCubicCrop(1.0)(cloud)
    torch.manual_seed(3)
    centre = cloud.xyz[torch.randint(0, cloud.xyz.shape[0], (1,))]
    inside = ((cloud.xyz >= centre - 0.5) & (cloud.xyz <= centre + 0.5)).all(1)
    assert 0 < len(crop) < len(cloud) and torch.equal(crop.xyz, cloud.xyz[inside]) and torch.equal(crop.class_l, cloud.class_l[inside])
    assert crop.medial_vector is not None  # filter keeps the labels (cloud.py:72-95) ...
    torch.manual_seed(5)
    scaled = aug.Scale(0.5, 0.6)(cloud)
    ratio = (scaled.xyz[10] / cloud.xyz[10]).cpu()
    assert scaled.medial_vector is None and 0.5 <= ratio[0] < 0.6 and torch.allclose(ratio, ratio[0].expand(3))  # ... scale drops them
    rot = aug.FixedRotate([np.pi / 2, 0.0, 0.0])(cloud)
    expect = torch.stack([cloud.xyz[:, 0], cloud.xyz[:, 2], -cloud.xyz[:, 1]], 1)  # xyz @ R_x(90 deg)
    assert torch.allclose(rot.xyz, expect, atol=1e-5)
    moved = aug.FixedTranslate([1.0, 2.0, 3.0])(cloud)
    assert torch.allclose(moved.xyz - cloud.xyz, torch.tensor([1.0, 2.0, 3.0], device=backend).expand_as(cloud.xyz), atol=1e-5)
    torch.manual_seed(1)
    drop = aug.RandomDropout(0.5)(cloud)
    assert 0.5 * len(cloud) <= len(drop) <= len(cloud)
    assert len(aug.RandomCrop(1.0, 1.0, 1.0)(cloud)) <= len(cloud)
    down = aug.VoxelDownsample(0.1)(cloud)
    q = torch.div(cloud.xyz, 0.1, rounding_mode="floor")
    assert len(down) == torch.unique(q, dim=0).shape[0] - 1  # the reference's off-by-one: the first voxel is skipped
    pipe = aug.AugmentationPipeline([aug.CentreCloud(), aug.FixedTranslate([0.0, 1.0, 0.0])])
    assert abs(pipe(cloud).xyz[:, 1].min().item() - 1.0) < 1e-5


def test_evaluate_losses_is_forward_plus_losses(backend, tmp_path):
    """Two labelled clouds through TreeDataset -> batch_collate -> the HIP network -> the fused losses; the mean losses equal
    the float64 oracle network + oracle losses on the same voxels (1e-3: the network's float32 accumulation order)."""
    from test_unet import random_state_dict

    clouds = [sample_tree_cloud(6000, seed=s, scale=0.6, max_depth=3, foliage_fraction=0.3) for s in (1, 2)]
    split = _write_split(tmp_path, clouds)
    ds = TreeDataset(0.05, split, tmp_path, "validation", ["xyz"], ["radius", "direction", "class_l"], device=backend)
    loader = [batch_collate([ds[0], ds[1]])]
    w = random_state_dict(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), seed=1)
    net = Smart_Tree(w, device=backend)
    net.use_mfma = backend.type != "cpu"  # (sanitizer build: the vector kernels; the matrix-core kernels have their tests in test_unet.py)
    fn = lambda p, t, m: L.compute_loss(p, t, m, L.L1Loss, L.cosine_similarity_loss, L.focal_loss, vector_class=0)
    got = L.evaluate_losses(loader, net, fn, device=backend)
    (inputs, targets), coords, mask, _ = loader[0]
    ref_preds = uo.OracleNet(w, dtype=torch.float64).forward(inputs.cpu().numpy(), coords.cpu().numpy())
    ref = lo.compute_loss(ref_preds, targets.cpu().numpy(), mask=mask.cpu().numpy(), vector_class=0)
    np.testing.assert_allclose([got["radius"], got["direction"], got["class_l"]],
                               [ref["radius"], ref["direction"], ref["class_l"]], rtol=1e-3)
    assert np.isclose(got["total"], got["radius"] + got["direction"] + got["class_l"])
