"""Cloud / skeleton file formats: .npz keys, PLY round trips, CLI overrides."""
from pathlib import Path

import numpy as np
import pytest
import torch

from smart_tree_amd import cli
from smart_tree_amd.data_types.branch import BranchSkeleton
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.data_types.tree import DisjointTreeSkeleton, TreeSkeleton
from smart_tree_amd.util import file as F


def test_npz_cloud_roundtrip_and_legacy_key(tmp_path):
    rng = np.random.RandomState(0)
    xyz, mv = rng.rand(50, 3).astype(np.float32), rng.rand(50, 3).astype(np.float32)
    np.savez(tmp_path / "legacy.npz", xyz=xyz, rgb=xyz, vector=mv, class_l=np.zeros((50, 1)))
    c = F.load_cloud(tmp_path / "legacy.npz")
    assert torch.equal(c.medial_vector, torch.from_numpy(mv)) and c.class_l.shape == (50, 1) and c.filename.name == "legacy.npz"
    F.save_cloud(tmp_path / "again.npz", c)
    c2 = F.load_cloud(tmp_path / "again.npz")
    assert torch.equal(c2.xyz, c.xyz) and torch.equal(c2.medial_vector, c.medial_vector)


def test_ply_points_binary_and_ascii(tmp_path):
    rng = np.random.RandomState(1)
    xyz, rgb = rng.rand(20, 3).astype(np.float32), (rng.randint(0, 256, (20, 3)) / 255.0).astype(np.float32)
    F.write_ply_points(tmp_path / "c.ply", xyz, rgb)
    c = F.load_cloud(tmp_path / "c.ply")
    np.testing.assert_array_equal(c.xyz.numpy(), xyz)
    np.testing.assert_allclose(c.rgb.numpy(), rgb, atol=1 / 255)
    lines = ["ply", "format ascii 1.0", "element vertex 3", "property float x", "property float y", "property float z",
             "end_header", "0 1 2", "3 4 5", "6 7 8"]
    (tmp_path / "a.ply").write_text("\n".join(lines) + "\n")
    xyz2, rgb2 = F.read_ply_points(tmp_path / "a.ply")
    assert rgb2 is None and xyz2.tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def test_skeleton_outputs(tmp_path):
    b0 = BranchSkeleton(0, -1, torch.rand(4, 3), torch.rand(4, 1))
    b1 = BranchSkeleton(1, 0, torch.rand(3, 3), torch.rand(3, 1))
    b1.radii = b1.radii.reshape(-1)  # smoothed branches carry 1-D radii (tree.py:130-134)
    sk = DisjointTreeSkeleton([TreeSkeleton(0, {0: b0, 1: b1})])
    F.save_skeleton_npz(tmp_path / "s.npz", sk)
    z = np.load(tmp_path / "s.npz")
    assert z["branches"].tolist() == [[0, 0, -1, 0, 4], [0, 1, 0, 4, 3]] and z["xyz"].shape == (7, 3)
    F.write_ply_skeleton(tmp_path / "s.ply", sk)
    head = (tmp_path / "s.ply").read_bytes().split(b"end_header")[0].decode()
    assert "element vertex 7" in head and "element edge 5" in head


def test_cli_overrides():
    cfg = cli.load_config(["+path=tree.npz", "pipeline.skeletonizer.K=8", "pipeline.repair_skeletons=False"])
    assert cfg["path"] == "tree.npz" and cfg["pipeline"]["skeletonizer"]["K"] == 8
    assert cfg["pipeline"]["repair_skeletons"] is False and cfg["pipeline"]["model_inference"]["block_size"] == 4


def test_skeleton_npz_is_the_reference_layout(tmp_path):
    """tests/golden/ref_saved_skeleton.npz was written by the reference's own save_skeleton (tools/make_goldens.py):
    our load_skeleton reads it, and our save_skeleton writes the same keys, shapes, dtypes and values."""
    from smart_tree_amd.util.file import load_skeleton, save_skeleton

    gold = Path(__file__).resolve().parent / "golden"
    ref_file = gold / "ref_saved_skeleton.npz"
    tree = load_skeleton(ref_file)
    g = np.load(gold / "skeleton_y_tree.npz")
    assert list(tree.branches) == g["branch_ids"].tolist()
    for k, par in zip(g["branch_ids"].tolist(), g["branch_parent"].tolist()):
        b = tree.branches[k]
        assert b.parent_id == par
        np.testing.assert_array_equal(b.xyz.numpy(), g[f"branch_{k}_xyz"])
        np.testing.assert_array_equal(b.radii.numpy().reshape(-1), g[f"branch_{k}_radii"].reshape(-1))
    # write what the reference wrote: rebuild the tree it saved (radii [m,1], tree id 3)
    for k in tree.branches:
        tree.branches[k].radii = torch.from_numpy(g[f"branch_{k}_radii"])
    tree._id = 3
    save_skeleton(tree, tmp_path / "mine.npz")
    with np.load(ref_file) as a, np.load(tmp_path / "mine.npz") as b:
        assert sorted(a.files) == sorted(b.files)
        for key in a.files:
            assert a[key].shape == b[key].shape and a[key].dtype == b[key].dtype, key
            np.testing.assert_array_equal(a[key], b[key])


def test_cli_directory_runs_every_cloud_file(tmp_path, monkeypatch):
    """`+directory=` walks every entry (reference cli.py:22-23), .npz and .ply alike; other files are named and skipped."""
    from smart_tree_amd import cli

    seen = []

    class FakePipeline:
        def process_cloud(self, path):
            seen.append(Path(path).name)

    monkeypatch.setattr(cli, "instantiate", lambda node: FakePipeline())
    for name in ("b.npz", "a.ply", "notes.txt"):
        (tmp_path / name).write_bytes(b"")
    cli.main([f"+directory={tmp_path}"])
    assert seen == ["a.ply", "b.npz"]


def test_tube_mesh_matches_the_reference_functions(tmp_path):
    """tests/golden/tube_mesh.npz: the reference's own tube_vertices / cylinder_triangles (geometries.py:157-189) on a golden
    branch with the random start vector pinned -- ring vertices and the triangle list are reproduced exactly."""
    from smart_tree_amd.util import mesh

    g = np.load(Path(__file__).resolve().parent / "golden" / "tube_mesh.npz")
    v = mesh.tube_vertices(g["points"], g["radii"], 10, g["start"])
    np.testing.assert_array_equal(v, g["vertices"])
    np.testing.assert_array_equal(mesh.cylinder_triangles(v.shape[1], v.shape[0]), g["triangles"])
    verts, tris = mesh.tube_mesh(g["points"], g["radii"], 10, g["start"])
    assert verts.shape == (400, 3) and tris.max() == 399 and tris.shape == (780, 3)
    # ring vertices sit at the branch radius from the axis
    d = np.linalg.norm(v - g["points"][:, None, :], axis=2)
    np.testing.assert_allclose(d, np.broadcast_to(g["radii"][:, None], d.shape), rtol=1e-5)
    # merged mesh of a two-branch skeleton + PLY writer
    b0 = BranchSkeleton(0, -1, torch.from_numpy(g["points"]), torch.from_numpy(g["radii"]).reshape(-1, 1))
    b1 = BranchSkeleton(1, 0, torch.from_numpy(g["points"][:5] + 1.0), torch.from_numpy(g["radii"][:5]).reshape(-1, 1))
    sk = DisjointTreeSkeleton([TreeSkeleton(0, {0: b0, 1: b1})])
    mv, mt = mesh.skeleton_mesh(sk, start=g["start"])
    assert mv.shape == (450, 3) and mt.shape == (780 + 80, 3) and mt[780:].min() == 400
    mesh.write_ply_mesh(tmp_path / "mesh.ply", mv, mt)
    head = (tmp_path / "mesh.ply").read_bytes()[:200].decode("ascii", "replace")
    assert "element vertex 450" in head and "element face 860" in head


@pytest.mark.gpu
def test_cli_file_route_on_the_gpu(tmp_path):
    """SURVEY 8f.1 end to end on the MI355X: a synthetic cloud written as `.npz` and as `.ply`, `cli.main([...])` with
    `+path=` / `+directory=` and `save_outputs` (reference cli.py:18-26, pipeline.py:85-93) through the HIP path, and the files
    `save_outputs` wrote read back: `skeleton_0.npz` (the reference's own per-tree layout, util/file.py:73-116) equals
    `process_cloud(cloud=...)` of the same cloud, branch by branch; `cloud.ply` holds the labelled cloud."""
    from smart_tree_amd import cli
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.synthetic import sample_tree_cloud
    from smart_tree_amd.util.file import load_cloud, load_skeleton, read_ply_points, write_ply_points

    assert torch.cuda.is_available()
    c = sample_tree_cloud(120_000, seed=4, scale=0.8, max_depth=5)
    np.savez(tmp_path / "tree.npz", xyz=c["xyz"], rgb=c["rgb"])
    write_ply_points(tmp_path / "tree.ply", c["xyz"], c["rgb"])
    common = ["pipeline.save_outputs=True", "pipeline.model_inference.voxel_size=0.02"]  # devices default to cuda:0, as in the reference
    out_npz, out_ply = tmp_path / "out_npz", tmp_path / "out_ply"
    for src, out in ((tmp_path / "tree.npz", out_npz), (tmp_path / "tree.ply", out_ply)):
        out.mkdir()
        cli.main([f"+path={src}", f"pipeline.save_path={out}"] + common)
        for name in ("skeleton.npz", "skeleton_0.npz", "skeleton.ply", "mesh.ply", "cloud.ply"):
            assert (out / name).stat().st_size > 0, name
    # the same cloud through process_cloud(cloud=...)
    cfg = cli.load_config(common)
    pipe = cli.instantiate(cfg["pipeline"])
    pipe.save_outputs = False
    direct = pipe.process_cloud(cloud=load_cloud(tmp_path / "tree.npz"))
    tree0 = direct.skeletons[0]
    assert len(tree0.branches) >= 3
    for out in (out_npz, out_ply):  # the .ply route carries float32 xyz exactly; its rgb is quantised to 8 bit (not a network input)
        back = load_skeleton(out / "skeleton_0.npz")
        assert list(back.branches) == list(tree0.branches)
        for k, b in tree0.branches.items():
            assert back.branches[k].parent_id == b.parent_id
            np.testing.assert_array_equal(back.branches[k].xyz.numpy(), b.xyz.numpy())
            np.testing.assert_array_equal(back.branches[k].radii.numpy().reshape(-1), b.radii.numpy().reshape(-1))
        xyz, _ = read_ply_points(out / "cloud.ply")
        np.testing.assert_array_equal(xyz, pipe.last_labelled_cloud.xyz.cpu().numpy())
    # +directory= over both files (reference cli.py:22-23)
    both = tmp_path / "out_dir"
    both.mkdir()
    cli.main([f"+directory={tmp_path}", f"pipeline.save_path={both}"] + common)
    assert (both / "skeleton_0.npz").stat().st_size > 0
