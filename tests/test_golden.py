"""Golden vectors produced by the REFERENCE's own glue code (tools/make_goldens.py, build container
only): they pin the oracle, and the HIP path is checked against them directly as well."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import skeleton_oracle as so
from oracle import voxel_oracle as vo
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

GOLD = Path(__file__).resolve().parent / "golden"
CASES = ["skeleton_y_tree", "skeleton_small_tree"]


def _radius(mv):
    return np.sqrt(((mv * mv)[:, 0] + (mv * mv)[:, 1]) + (mv * mv)[:, 2]).astype(np.float32)


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_glue(case):
    g = np.load(GOLD / f"{case}.npz")
    xyz, mv = g["raw_xyz"], g["raw_medial_vector"]
    keep = so.outlier_removal(xyz + mv, _radius(mv), 8)
    np.testing.assert_array_equal(keep, g["keep_mask"])  # reference outlier_removal
    xyz, mv = xyz[keep], mv[keep]
    medial, radius = xyz + mv, _radius(mv)
    edges, w = so.nn_graph(medial, np.maximum(radius, np.float32(0.02)), 16)
    np.testing.assert_array_equal(edges, g["edges"])  # reference make_edges incl. the idx > 0 quirk
    np.testing.assert_array_equal(w, g["weights"])
    ids = g["component"]
    branches, _, _ = so.sample_tree(medial[ids], radius[ids], g["preds"], g["dist"])  # reference sample_tree's inputs
    assert [b.branch_id for b in branches] == g["branch_ids"].tolist()
    assert [b.parent_id for b in branches] == g["branch_parent"].tolist()
    for b in branches:
        np.testing.assert_array_equal(medial[ids][b.verts], g[f"branch_{b.branch_id}_xyz"])
        np.testing.assert_array_equal(radius[ids][b.verts].reshape(-1, 1), g[f"branch_{b.branch_id}_radii"])
    # post-processing: same branches survive; floats agree to float32 round-off (the reference's
    # einsum / conv1d summation order is torch's, the oracle's is explicit)
    tree = po.OTree(0, {b.branch_id: po.OBranch(b.branch_id, b.parent_id, medial[ids][b.verts],
                                               radius[ids][b.verts].reshape(-1, 1)) for b in branches})
    po.post_process([tree], True, 0.01, 0.02, True, True, 11)
    assert list(tree.branches) == g["post_ids"].tolist()
    for k, b in tree.branches.items():
        np.testing.assert_allclose(b.xyz, g[f"post_{k}_xyz"], rtol=1e-5, atol=1e-6)
        assert b.radii.shape == g[f"post_{k}_radii"].shape
        np.testing.assert_allclose(b.radii, g[f"post_{k}_radii"], rtol=1e-5, atol=1e-7)


def test_oracle_blocking_matches_reference_glue():
    g = np.load(GOLD / "blocking_50k.npz")
    c = sample_tree_cloud(50_000, seed=0)
    xyz = vo.centre_cloud(c["xyz"])
    np.testing.assert_array_equal(xyz, g["centred_xyz"])  # CentreCloud
    _, centres = vo.compute_blocks(xyz, 4.0, 20)
    np.testing.assert_array_equal(centres, g["block_centres"])  # compute_blocks: unique + count > 20
    for i, ctr in enumerate(centres):
        member = vo.cube_mask(xyz, ctr, 4.0 + 0.4 * 2)
        assert member.sum() == g["block_sizes"][i]
        np.testing.assert_array_equal(xyz[member][:64], g[f"block_{i}_first_xyz"])
        assert vo.cube_mask(xyz[member], ctr, 4.0).sum() == g[f"block_{i}_inner_count"]
    # the whole collated batch the reference's __getitem__ / batch_collate emit (inner mask on the representatives)
    vx = vo.voxelize_cloud(xyz, c["rgb"], 0.02)
    np.testing.assert_array_equal(vx["coords"], g["collated_coords"])
    np.testing.assert_array_equal(vx["mask"], g["collated_mask"])
    np.testing.assert_array_equal(vx["feats"][:, :3], g["collated_xyz"])


@pytest.mark.parametrize("case", CASES)
def test_hip_skeletonizer_matches_reference_glue(backend, case):
    """HIP path -> the branches the reference's own sample_tree / prune / repair / smooth produced."""
    if backend.type == "cpu" and case != "skeleton_y_tree":
        pytest.skip("the CPU sanitizer build is too slow for this case; it runs on the GPU")
    g = np.load(GOLD / f"{case}.npz")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=backend)
    sk.block_threads = 128 if backend.type == "cpu" else 0
    out = sk.forward(Cloud(xyz=t(g["raw_xyz"]), medial_vector=t(g["raw_medial_vector"])))
    out.prune(min_radius=0.01, min_length=0.02)
    out.repair()
    out.smooth(11)
    tree = out.skeletons[0]  # largest component = the one the golden holds
    assert list(tree.branches) == g["post_ids"].tolist()
    for k, b in tree.branches.items():
        np.testing.assert_allclose(b.xyz.numpy(), g[f"post_{k}_xyz"], rtol=1e-5, atol=1e-6)
        assert tuple(b.radii.shape) == g[f"post_{k}_radii"].shape
        np.testing.assert_allclose(b.radii.numpy(), g[f"post_{k}_radii"], rtol=1e-5, atol=1e-7)
    # and un-post-processed: exact
    raw = sk.forward(Cloud(xyz=t(g["raw_xyz"]), medial_vector=t(g["raw_medial_vector"]))).skeletons[0]
    assert list(raw.branches) == g["branch_ids"].tolist()
    for k, b in raw.branches.items():
        assert b.parent_id == int(g["branch_parent"][list(raw.branches).index(k)])
        np.testing.assert_array_equal(b.xyz.numpy(), g[f"branch_{k}_xyz"])
        np.testing.assert_array_equal(b.radii.numpy(), g[f"branch_{k}_radii"])


def test_hip_blocking_matches_reference_glue(backend):
    g = np.load(GOLD / "blocking_50k.npz")
    xyz = torch.from_numpy(g["centred_xyz"]).to(backend)
    out = voxelize_blocks(xyz, None, 0.02)
    np.testing.assert_array_equal(out.block_centres.cpu().numpy(), g["block_centres"])
    # the reference's own __getitem__ + batch_collate (PointToVoxel stand-in, tools/make_goldens.py): equality
    np.testing.assert_array_equal(out.coords.cpu().numpy(), g["collated_coords"])
    np.testing.assert_array_equal(out.mask.cpu().numpy(), g["collated_mask"])
    np.testing.assert_array_equal(out.feats[:, :3].cpu().numpy(), g["collated_xyz"])


@pytest.mark.parametrize("kernel", [4, 7])
def test_device_smooth_matches_conv1d_same_padding_for_even_kernels(backend, kernel):
    """tree.py:123-134 smooths with F.conv1d(padding="same"): an even kernel pads (k-1)//2 on the left and the extra
    sample on the right.  The deferred device box filter, the host fallback (F.conv1d itself) and the oracle agree."""
    g = np.load(GOLD / "skeleton_y_tree.npz")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=backend)
    sk.block_threads = 128 if backend.type == "cpu" else 0
    cloud = Cloud(xyz=t(g["raw_xyz"]), medial_vector=t(g["raw_medial_vector"]))
    dev_out = sk.forward(cloud)
    dev_out.smooth(kernel)  # deferred: k_post_process
    host_out = sk.forward(cloud)
    _ = host_out.skeletons  # materialise first -> TreeSkeleton.smooth = F.conv1d(padding="same")
    host_out.smooth(kernel)
    n_smoothed = 0
    for td, th in zip(dev_out.skeletons, host_out.skeletons):
        assert list(td.branches) == list(th.branches)
        for k in td.branches:
            rd, rh = td.branches[k].radii.numpy(), th.branches[k].radii.numpy()
            assert rd.shape == rh.shape
            np.testing.assert_allclose(rd, rh, rtol=1e-5, atol=1e-7)
            n_smoothed += rd.ndim == 1
        ot = po.OTree(0, {k: po.OBranch(k, b.parent_id, b.xyz.numpy(), b.radii.numpy()) for k, b in
                          sk.forward(cloud).skeletons[td._id].branches.items()})
        po.smooth(ot, kernel)
        for k in td.branches:
            np.testing.assert_array_equal(td.branches[k].radii.numpy(), ot.branches[k].radii)
    assert n_smoothed >= 2
    # the advisor's example: k = 4 on [1,2,4,8,16,32] -> [7,15,30,60,56,48] / 4
    ob = po.OTree(0, {0: po.OBranch(0, -1, np.zeros((6, 3), np.float32), np.array([1, 2, 4, 8, 16, 32], np.float32))})
    po.smooth(ob, 4)
    np.testing.assert_array_equal(ob.branches[0].radii, np.array([7, 15, 30, 60, 56, 48], np.float32) / 4)


# ---- the reference's WHOLE Skeletonizer.forward (tools/make_goldens.py::quirks_case), every SURVEY 8c(4) quirk asserted to fire there:
# vertex 0's in-edges dropped (graph.py:59), single-vertex paths consumed without a branch (path.py:125-126), the root-reaching
# path's parent read from branch_ids[-1] (path.py:132), components of 31 / 32 vertices (data_types/graph.py:44-45), size order
def _quirk_trees(g):
    for ti in range(int(g["n_trees"])):
        ids = g[f"tree_{ti}_ids"].tolist()
        yield ti, ids, g[f"tree_{ti}_parent"].tolist(), [g[f"tree_{ti}_branch_{k}_xyz"] for k in ids], [g[f"tree_{ti}_branch_{k}_radii"] for k in ids]


def test_oracle_matches_reference_forward_with_every_quirk():
    g = np.load(GOLD / "skeleton_quirks.npz")
    assert int(g["single_vertex_paths"]) >= 1 and 32 in g["component_sizes"] and 31 not in g["component_sizes"]
    xyz, mv = g["raw_xyz"], g["raw_medial_vector"]
    ref = so.skeletonize(xyz, mv)
    np.testing.assert_array_equal(ref.keep_mask, g["keep_mask"])
    np.testing.assert_array_equal(ref.edges, g["edges"])
    np.testing.assert_array_equal(ref.weights, g["weights"])
    assert not (ref.edges[:, 1] == 0).any()  # graph.py:59
    sizes = np.unique(ref.labels, return_counts=True)[1]
    assert 31 in sizes  # a 31-vertex component exists in the graph and is dropped
    assert [len(c.vertex_ids) for c in ref.components] == g["component_sizes"].tolist()
    medial, radius = (xyz + mv)[ref.keep_mask], _radius(mv)[ref.keep_mask]
    iterations = 0
    for (ti, ids, parents, pts, radii), comp in zip(_quirk_trees(g), ref.components):
        np.testing.assert_array_equal(comp.preds, g[f"tree_{ti}_preds"])  # the stand-in SSSP is the oracle's: pins the renumbering around it
        np.testing.assert_array_equal(comp.tree_dist, g[f"tree_{ti}_dist"])  # reference pred_graph + second sssp
        np.testing.assert_array_equal(comp.dist, comp.tree_dist)
        assert [b.branch_id for b in comp.branches] == ids and [b.parent_id for b in comp.branches] == parents
        for b, p, r in zip(comp.branches, pts, radii):
            np.testing.assert_array_equal(medial[comp.vertex_ids][b.verts], p)
            np.testing.assert_array_equal(radius[comp.vertex_ids][b.verts].reshape(-1, 1), r)
        iterations += so.sample_tree(medial[comp.vertex_ids], radius[comp.vertex_ids], comp.preds, comp.tree_dist)[2]
    assert iterations == int(g["iterations"])  # incl. the single-vertex paths that consume points without a branch


def test_hip_matches_reference_forward_with_every_quirk(backend):
    g = np.load(GOLD / "skeleton_quirks.npz")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=backend)
    sk.block_threads = 128 if backend.type == "cpu" else 0
    out = sk.forward(Cloud(xyz=t(g["raw_xyz"]), medial_vector=t(g["raw_medial_vector"])))
    assert len(out.skeletons) == int(g["n_trees"])
    for (ti, ids, parents, pts, radii), tree in zip(_quirk_trees(g), out.skeletons):
        assert list(tree.branches) == ids
        assert [b.parent_id for b in tree.branches.values()] == parents
        for b, p, r in zip(tree.branches.values(), pts, radii):
            np.testing.assert_array_equal(b.xyz.numpy(), p)
            np.testing.assert_array_equal(b.radii.numpy(), r)
