"""st_voxelize_blocks vs oracle/voxel_oracle.py: bit-exact coordinates, order, masks, features."""
import numpy as np
import pytest
import torch

from oracle import voxel_oracle as vo
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.synthetic import sample_tree_cloud


def _compare(xyz, rgb, voxel_size, device, **kw):
    ref = vo.voxelize_cloud(xyz, rgb, voxel_size, **kw)
    out = voxelize_blocks(torch.from_numpy(xyz).to(device), torch.from_numpy(rgb).to(device), voxel_size, **kw)
    assert out.coords.shape[0] == ref["coords"].shape[0]
    np.testing.assert_array_equal(out.block_centres.cpu().numpy(), ref["centres"])
    np.testing.assert_array_equal(out.coords.cpu().numpy(), ref["coords"])
    np.testing.assert_array_equal(out.point_index.cpu().numpy(), ref["point"])
    np.testing.assert_array_equal(out.mask.cpu().numpy(), ref["mask"])
    np.testing.assert_array_equal(out.feats.cpu().numpy(), ref["feats"])
    return out


@pytest.mark.parametrize("n,voxel", [(4000, 0.02), (20000, 0.05)])
def test_voxelize_tree(backend, n, voxel):
    c = sample_tree_cloud(n, seed=3)
    xyz = vo.centre_cloud(c["xyz"])
    rgb = np.random.RandomState(0).rand(n, 3).astype(np.float32)
    out = _compare(xyz, rgb, voxel, backend)
    assert out.coords.shape[0] > 0


@pytest.mark.parametrize("block,buffer,voxel,min_points", [(3.0, 0.25, 0.03, 20), (0.7, 0.1, 0.013, 5), (8.0, 0.4, 0.1, 50), (2.5, 1.2, 0.04, 1)])
def test_voxelize_other_block_geometries(backend, block, buffer, voxel, min_points):
    """Block sizes that are not powers of two (the kernel divides instead of multiplying by the reciprocal), voxel sizes that do not
    divide the block, other min_points: conf/tree-dataset.yaml's keywords away from the shipped values."""
    c = sample_tree_cloud(9000, seed=5, scale=0.8)
    xyz = vo.centre_cloud(c["xyz"])
    out = _compare(xyz, np.zeros_like(xyz), voxel, backend, block_size=block, buffer_size=buffer, min_points=min_points)
    assert out.coords.shape[0] > 100 and out.block_centres.shape[0] >= 2


def test_voxelize_block_boundaries(backend):
    """Points exactly on block / halo faces, duplicates, a block below the min_points threshold."""
    rng = np.random.RandomState(1)
    dense = rng.uniform(-0.5, 4.5, (3000, 3)).astype(np.float32)
    faces = np.array([[0, 0, 0], [4, 4, 4], [3.6, 1, 1], [4.4, 1, 1], [-0.4, 2, 2], [0.4, 2, 2],
                      [2.0, 2.0, 2.0], [2.0, 2.0, 2.0], [4.0, 0.0, 3.999999]], np.float32)
    sparse = rng.uniform(8.1, 9.0, (15, 3)).astype(np.float32)  # own block, <= 20 points: vanishes
    xyz = np.concatenate([dense, faces, sparse])
    rgb = np.zeros_like(xyz)
    _compare(xyz, rgb, 0.1, backend)


def test_voxelize_empty_result(backend):
    xyz = np.random.RandomState(2).uniform(0, 3, (10, 3)).astype(np.float32)  # no block has > 20 points
    out = _compare(xyz, np.zeros_like(xyz), 0.02, backend)
    assert out.coords.shape[0] == 0 and out.block_centres.shape[0] == 0


def test_voxelize_sparse_cloud_takes_the_capacity_retry(backend):
    """More voxels than the first capacity guess (half a voxel per point): points farther apart than a voxel, each sitting in
    up to eight halo cubes.  The hash table fills, inserts give up after the probe limit, the kernel flags it, the host retries
    with the worst-case sizes -- and the next call with the same parameters starts from the ratio it has seen."""
    from smart_tree_amd.dataset import dataset as ds

    rng = np.random.RandomState(5)
    xyz = rng.uniform(-0.3, 0.3, (6000, 3)).astype(np.float32)  # around a block corner: every point is in 8 halo cubes
    kw = dict(block_size=1.0, buffer_size=0.4)
    ds._VOXELS_PER_POINT.pop((0.001, 1.0, 0.4), None)
    out = _compare(xyz, np.zeros_like(xyz), 0.001, backend, **kw)
    assert out.coords.shape[0] > 3 * xyz.shape[0]  # > the second guess as well: the 8n capacity was needed
    assert ds._VOXELS_PER_POINT[(0.001, 1.0, 0.4)] > 3
    _compare(xyz, np.zeros_like(xyz), 0.001, backend, **kw)  # second call: first guess already large enough


def test_voxelize_halo_wider_than_half_a_block(backend):
    """buffer >= block / 2: a point can sit in all three block columns of an axis (up to 27 halo cubes) -- the general walk,
    not the eight-block fast path."""
    rng = np.random.RandomState(8)
    xyz = rng.uniform(-1.5, 1.5, (5000, 3)).astype(np.float32)
    out = _compare(xyz, np.zeros_like(xyz), 0.05, backend, block_size=1.0, buffer_size=0.6)
    assert out.block_centres.shape[0] >= 27 and out.coords.shape[0] > 5 * xyz.shape[0]


def test_voxelize_plot_larger_than_the_default_block_table(backend):
    """A cloud whose block-id bounding box has more cells than the default table (32768): the reference's torch.unique-based
    compute_blocks takes any extent; here the call is retried with a larger table (advisor, round 1)."""
    rng = np.random.RandomState(4)
    a = rng.uniform(0, 2, (400, 3)).astype(np.float32)
    far = (a[:300] + np.array([150.0, 10.0, 170.0], np.float32)).astype(np.float32)  # 38 x 3 x 43 blocks of 4 m... x 8 = > 32768 cells
    xyz = np.concatenate([a, far, a[:50] + np.array([0.0, 130.0, 0.0], np.float32)])
    out = _compare(xyz, np.zeros_like(xyz), 0.1, backend)
    assert out.block_centres.shape[0] >= 2


def test_whole_cloud_grid_beyond_the_key_range_is_refused(backend):
    """whole-cloud mode turns a cloud's bounding box into ONE voxel grid; the hash key gives each axis 16 bits,
    so a span of more than 65535 cells (here 70 m at 1 mm) must fail loudly instead of merging distinct voxels."""
    from smart_tree_amd import _lib
    from smart_tree_amd.dataset.dataset import voxelize_cloud

    xyz = torch.tensor([[0.0, 0.0, 0.0], [0.5, 0.2, 0.1], [70.0, 1.0, 1.0]], device=backend)
    with pytest.raises(_lib.StError, match="65535 cells"):
        voxelize_cloud(xyz, None, 0.001)
    ok = voxelize_cloud(xyz, None, 0.01)  # 7000 cells: fine
    assert ok.coords.shape[0] == 2  # (the point ON the maximum face has no cell: PointToVoxel drops c == grid)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_non_finite_coordinates_are_refused(backend, bad):
    """A NaN / infinite coordinate has no block (the reference's torch.unique over block ids would group it somewhere arbitrary):
    the call fails and says why, for one cloud and inside a batch."""
    c = sample_tree_cloud(4000, seed=2, scale=0.5, max_depth=3)
    xyz = c["xyz"].copy()
    xyz[900, 1] = bad
    with pytest.raises(RuntimeError, match="non-finite"):
        voxelize_blocks(torch.from_numpy(xyz).to(backend), None, 0.02)
    both = torch.from_numpy(np.concatenate([c["xyz"], xyz])).to(backend)
    seg = torch.tensor([0, 4000, 8000], dtype=torch.int32, device=both.device)
    with pytest.raises(RuntimeError, match="non-finite"):
        voxelize_blocks(both, None, 0.02, seg_off=seg)
