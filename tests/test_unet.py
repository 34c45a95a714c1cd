"""Rulebooks (bit-exact) and network outputs (fp32 tolerance) of the HIP path vs oracle/unet_oracle.py."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import unet_oracle as uo
from oracle import voxel_oracle as vo
from smart_tree_amd.model import sparse_ops as ops
from smart_tree_amd.model.model import Smart_Tree
from smart_tree_amd.model.sparse import sparse_from_batch
from smart_tree_amd.synthetic import sample_tree_cloud

WEIGHTS = Path(__file__).resolve().parents[1] / "smart_tree_amd" / "model" / "weights"


def _small_batch(n=6000, voxel=0.05, seed=5):
    c = sample_tree_cloud(n, seed=seed, scale=0.6, max_depth=4)
    xyz = vo.centre_cloud(c["xyz"])
    return vo.voxelize_cloud(xyz, c["rgb"], voxel, block_size=2.0, buffer_size=0.2)


def test_sparse_tensor_attributes_of_the_reference():
    """sparse.py:9-19 of the reference: spatial_shape = the largest z / y / x (not + 1), batch_size = number of voxels.
    spatial_shape is reduced on first access (no kernel reads it) and travels through replace_feature."""
    coords = torch.tensor([[0, 3, 9, 1], [1, 7, 2, 4], [0, 0, 0, 11]], dtype=torch.int32)
    sp = sparse_from_batch(torch.zeros(3, 3), coords, torch.device("cpu"))
    assert sp.batch_size == 3 and sp._spatial_shape is None
    assert sp.spatial_shape.tolist() == [7, 9, 11] and sp.spatial_shape.dtype == torch.int32
    assert sp.replace_feature(torch.ones(3, 8)).spatial_shape.tolist() == [7, 9, 11]
    empty = sparse_from_batch(torch.zeros(0, 3), torch.zeros((0, 4), dtype=torch.int32), torch.device("cpu"))
    assert empty.spatial_shape.tolist() == [0, 0, 0]


def test_sparse_tensor_attributes_equal_the_reference_golden(backend):
    """tests/golden/sparse_attrs.npz is what the reference's OWN sparse_from_batch (model/sparse.py:9-19) handed to spconv for the
    collated batch of blocking_50k.npz (tools/make_goldens.py::sparse_attrs_case, a recorder in place of SparseConvTensor):
    spatial_shape = the largest z / y / x -- one LESS than the true extent --, batch_size = the number of voxels, int32 indices.
    The mirrored attributes must equal it on both backends; and the rulebooks must NOT take their extent from it (the documented
    canonical choice): the coarse set of the first strided conv is the one the oracle derives from the
    true extent max + 1, which differs from what the attribute's extent would give (max-face voxels keep their outputs)."""
    gold = Path(__file__).parent / "golden"
    g, a = np.load(gold / "blocking_50k.npz"), np.load(gold / "sparse_attrs.npz")
    coords = torch.from_numpy(g["collated_coords"])
    feats = torch.from_numpy(g["collated_xyz"])
    sp = sparse_from_batch(feats, coords.float(), backend)  # (batch_collate emits float coordinates, dataset.py:216)
    assert sp.indices.dtype == torch.int32 and str(sp.indices.dtype) == str(a["indices_dtype"])
    assert sp.indices.device.type == torch.device(backend).type
    assert sp.batch_size == int(a["batch_size"]) == coords.shape[0]
    assert sp.spatial_shape.cpu().tolist() == a["spatial_shape"].tolist()
    assert (a["spatial_shape"] + 1).tolist() == a["true_extent"].tolist()  # the quirk: the attribute is max, not max + 1
    np.testing.assert_array_equal(sp.indices.cpu().numpy(), g["collated_coords"])
    # the rulebook builders clip the coarse set at the TRUE extent: equal to the oracle's coarse set, and at least one coarse
    # voxel exists that an extent of `spatial_shape` (one less) would have cut off
    pyr = ops.build_pyramid(sp.indices, depth=1)
    coarse = uo.strided_out_coords(g["collated_coords"])
    np.testing.assert_array_equal(pyr.coords[1].cpu().numpy(), coarse)
    ext_attr = (a["spatial_shape"] - 1) // 2 + 1  # spconv's output-size formula (k 3, s 2, p 1) on the attribute's extent
    ext_true = (a["true_extent"] - 1) // 2 + 1
    assert (coarse[:, 1:].max(0) + 1 <= ext_true).all()
    assert (coarse[:, 1:].max(0) + 1 > ext_attr).any(), "this batch has no max-face voxel: the golden no longer exercises the choice"


def test_rulebooks_bit_exact(backend):
    vx = _small_batch()
    coords = torch.from_numpy(vx["coords"]).to(backend)
    pyr = ops.build_pyramid(coords, depth=3)
    level_coords = vx["coords"]
    for level in range(4):
        np.testing.assert_array_equal(pyr.coords[level].cpu().numpy(), level_coords)
        np.testing.assert_array_equal(pyr.subm[level].cpu().numpy(), uo.subm_rulebook(level_coords))
        if level == 3:
            break
        coarse = uo.strided_out_coords(level_coords)
        np.testing.assert_array_equal(pyr.down[level].cpu().numpy(), uo.down_rulebook(coarse, level_coords))
        np.testing.assert_array_equal(pyr.up[level].cpu().numpy(), uo.up_rulebook(level_coords, coarse))
        level_coords = coarse


def test_strided_rulebook_isolated_odd_voxels_take_the_retry_path(backend):
    """An isolated voxel with odd coordinates reaches up to 8 coarse outputs: n_out ~ 6.4 n overflows the first
    capacity guess (n + 1024), the build must notice (device-side flag, read after the last pass) and retry."""
    g = np.arange(7, dtype=np.int32) * 10 + 5
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    coords = np.stack([np.zeros(z.size, np.int32), z.ravel(), y.ravel(), x.ravel()], axis=1).astype(np.int32)
    h = ops.build_coord_hash(torch.from_numpy(coords).to(backend))
    rb = ops.build_strided_rulebook(torch.from_numpy(coords).to(backend), h)
    coarse = uo.strided_out_coords(coords)
    assert coarse.shape[0] > coords.shape[0] + 1024  # (2*7 - 1)^3: the max-face voxels lose the outputs beyond the extent
    np.testing.assert_array_equal(rb.out_coords.cpu().numpy(), coarse)
    np.testing.assert_array_equal(rb.nbr_down.cpu().numpy(), uo.down_rulebook(coarse, coords))
    np.testing.assert_array_equal(rb.nbr_up.cpu().numpy(), uo.up_rulebook(coords, coarse))


def test_up_order_groups_rows_by_parity_and_leaves_the_conv_unchanged(backend):
    """up_order: a permutation of the fine rows, grouped by coordinate parity class (z, y, x bits); launching the
    inverse conv in that order must give bit-identical features (every row is computed on its own)."""
    vx = _small_batch()
    coords = torch.from_numpy(vx["coords"]).to(backend)
    pyr = ops.build_pyramid(coords, depth=2)
    g = torch.Generator().manual_seed(3)
    for level, (cin, cout) in enumerate([(16, 8), (32, 16)]):
        fine = pyr.coords[level].cpu().numpy()
        tagged = pyr.up_order[level].cpu().numpy()
        order = tagged & 0x0FFFFFFF  # top four bits: 8 + parity class (lets the kernel skip the 19-26 dead offsets)
        assert sorted(order.tolist()) == list(range(fine.shape[0]))
        cls = ((fine[order, 1] & 1) << 2) | ((fine[order, 2] & 1) << 1) | (fine[order, 3] & 1)
        assert (np.diff(cls) >= 0).all()
        np.testing.assert_array_equal((tagged >> 28) & 0xF, 8 + cls)
        live = (pyr.up[level].cpu().numpy() >= 0)  # [27, n]: pairs exist only at the offsets the class allows
        for k in range(27):
            ok = np.ones(fine.shape[0], bool)
            for a, ka in enumerate((k // 9, (k // 3) % 3, k % 3)):
                ok &= ((fine[:, 1 + a] & 1) == 1) == (ka != 1)
            assert not (live[k] & ~ok).any()
        n_fine, n_coarse = fine.shape[0], pyr.coords[level + 1].shape[0]
        z = torch.randn((n_coarse, cin), generator=g).to(backend)
        w = (torch.randn((27, cin, cout), generator=g) * 0.1).to(backend)
        wp = ops.mfma_weight(w) if ops.mfma_eligible(cin, cout, cin) else None
        a = ops.sparse_conv(z, w, pyr.up[level], n_fine, relu=True, wp=wp)
        b = ops.sparse_conv(z, w, pyr.up[level], n_fine, relu=True, wp=wp, row_order=pyr.up_order[level])
        assert torch.equal(a, b)


def test_rulebook_empty(backend):
    coords = torch.zeros((0, 4), dtype=torch.int32, device=backend)
    pyr = ops.build_pyramid(coords, depth=3)
    assert all(c.shape[0] == 0 for c in pyr.coords)


def _tolerance_check(got, ref64, ref32, name):
    """fp32 parity bar: within 1e-4 relative of the fp64 oracle, measured against each output
    tensor's own scale (rms), and never worse than 4x the fp32 oracle's own distance from fp64
    (the checkpoints' BatchNorm statistics -- |mean| up to 4e3, var down to 1e-21 -- make ANY fp32
    evaluation order noisy at that level)."""
    scale = np.sqrt(np.mean(ref64 ** 2)) + 1e-30
    err = np.abs(got - ref64).max() / scale
    base = np.abs(ref32 - ref64).max() / scale
    assert err <= max(1e-4, 4 * base), f"{name}: rel err {err:.3e} (fp32 oracle itself {base:.3e})"


@pytest.mark.parametrize("ckpt", ["noble-elevator-58", "peach-forest-65"])
def test_unet_forward_matches_oracle(backend, ckpt):
    vx = _small_batch()
    w = uo.load_weights(WEIGHTS / f"{ckpt}.npz")
    ref64 = uo.OracleNet(w, dtype=torch.float64).forward(vx["feats"][:, :3], vx["coords"])
    ref32 = uo.OracleNet(w, dtype=torch.float32).forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w, device=backend)
    if backend.type == "cpu" and ckpt == "peach-forest-65":
        net.use_mfma = False  # sanitizer build: one checkpoint through the matrix-core kernels, the other through the vector kernels
    sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
    out = net.forward(sp)
    assert set(out) == {"radius", "direction", "class_l"}
    for k in out:
        assert out[k].shape == ref64[k].shape
        _tolerance_check(out[k].cpu().numpy().astype(np.float64), ref64[k], ref32[k].astype(np.float64), k)
    # fused inference tail: exp(radius) * direction, argmax
    r, d, c, mv, cls = net.forward_fused_tail(sp)
    mv_ref, cls_ref = uo.inference_tail(r.cpu().numpy(), d.cpu().numpy(), c.cpu().numpy())
    np.testing.assert_allclose(mv.cpu().numpy(), mv_ref, rtol=2e-6, atol=1e-30)
    np.testing.assert_array_equal(cls.cpu().numpy(), cls_ref)


def random_state_dict(template, seed=0):
    """Same keys/shapes as a checkpoint but well-conditioned random values, so every channel is
    alive (the shipped checkpoints saturate on small synthetic inputs and exercise little)."""
    rng = np.random.RandomState(seed)
    sd = {}
    for k, v in template.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v
        elif v.ndim == 5:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3] * v.shape[4]
            sd[k] = (rng.randn(*v.shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif k.endswith("running_var"):
            sd[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        elif k.endswith("running_mean"):
            sd[k] = (rng.randn(*v.shape) * 0.1).astype(np.float32)
        elif k.endswith(".weight"):
            sd[k] = rng.uniform(0.5, 1.5, v.shape).astype(np.float32)
        else:
            sd[k] = (rng.randn(*v.shape) * 0.1).astype(np.float32)
    return sd


def test_unet_random_weights_every_layer(backend):
    vx = _small_batch(n=5000, seed=9)
    w = random_state_dict(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), seed=1)
    oracle = uo.OracleNet(w, dtype=torch.float64)
    ref = oracle.forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w, device=backend)
    sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
    feats = net.features(sp).cpu().numpy()
    tail0 = oracle.trace["tail0"].numpy()
    assert (tail0 > 0).mean() > 0.2, "test input does not exercise the network"
    np.testing.assert_allclose(feats, tail0, rtol=1e-4, atol=1e-4 * np.abs(tail0).max())
    out = net.forward(sp)
    for k in out:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4 * np.abs(ref[k]).max())


def test_unet_of_the_training_config_depth(backend):
    """conf/training.yaml:123-127 trains `unet_planes: [8, 16, 32]` -- one level fewer than the shipped checkpoints.  The depth is read
    off the state dict; the three-level network (the deepest UBlock and the encoder / decoder / tail around it removed) must match
    the oracle as the four-level one does."""
    vx = _small_batch(n=5000, seed=9)
    w = random_state_dict(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), seed=1)
    gone = ("UNet.U.U.U.", "UNet.U.U.Encode", "UNet.U.U.Decode", "UNet.U.U.Tail")
    w3 = {k: v for k, v in w.items() if not k.startswith(gone)}
    assert len(w3) < len(w)
    ref = uo.OracleNet(w3, dtype=torch.float64).forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w3, device=backend)
    assert net.depth == 2
    out = net.forward(sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend))
    for k in out:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4 * np.abs(ref[k]).max())


@pytest.mark.parametrize("cin,cout,c0", [(6, 8, 6), (5, 7, 5), (24, 40, 12), (16, 48, 16), (128, 64, 64),
                                         (12, 6, 6), (20, 10, 10), (36, 18, 18), (7, 5, 3)])  # (concat splits that are no multiple of 4: planes 6 / 10 / 18)
def test_conv_of_any_channel_counts(backend, cin, cout, c0):
    """Channel counts no kernel is instantiated for (a model config away from the shipped planes; colour as input channels 4-6):
    the generic kernel behind st_sparse_conv_fwd, with concat, BatchNorm affine, residual, ReLU and a row order, against float64."""
    rng = np.random.RandomState(cin * 100 + cout)
    n, n_in, K = 517, 490, 27
    nbr = rng.randint(0, n_in, size=(K, n)).astype(np.int32)
    nbr[rng.rand(K, n) < 0.5] = -1
    x = rng.randn(n_in, cin).astype(np.float32)
    w = (rng.randn(K, cin, cout) * 0.1).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.randn(cout).astype(np.float32)
    res = rng.randn(n, cout).astype(np.float32)
    order = rng.permutation(n).astype(np.int32)
    ref = np.zeros((n, cout))
    for k in range(K):
        hit = nbr[k] >= 0
        ref[hit] += x[nbr[k][hit]].astype(np.float64) @ w[k].astype(np.float64)
    ref = np.maximum(ref * scale.astype(np.float64) + shift.astype(np.float64) + res.astype(np.float64), 0.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    x0, x1 = (t(x[:, :c0]), t(x[:, c0:])) if c0 < cin else (t(x), None)
    got = ops.sparse_conv(x0, t(w), t(nbr), n, x1=x1, scale=t(scale), shift=t(shift), residual=t(res), relu=True, row_order=t(order))
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    # pointwise (K = 1, no table), no epilogue
    y = ops.sparse_conv(t(x[:n]), t(w[:1]), None, min(n, n_in)).cpu().numpy()
    np.testing.assert_allclose(y, x[:min(n, n_in)].astype(np.float64) @ w[0].astype(np.float64), rtol=2e-5, atol=2e-5)


def _other_architecture(template, planes, hidden, n_classes):
    """The checkpoint's keys with other widths: unet_planes -> `planes`, head layers planes[0] -> hidden[0] -> hidden[1] -> outputs."""
    cmap = dict(zip((8, 16, 32, 64), planes))
    shapes = {}
    for k, v in template.items():
        name = k.split(".")[0]
        if name.endswith("_head"):
            i = int(k.split(".")[2])
            nout = {"radius_head": 1, "direction_head": 3, "class_head": n_classes}[name]
            widths = {0: (hidden[0], planes[0]), 1: (hidden[0],), 3: (hidden[1], hidden[0]), 4: (hidden[1],), 6: (nout, hidden[1])}[i]
            shapes[k] = (widths[0], 1, 1, 1, widths[1]) if v.ndim == 5 else (() if v.ndim == 0 else (widths[0],))
        elif v.ndim == 5 and (".Tail.sequence.0." in k or ".Tail.identity.0." in k):  # the concat of skip and decoded features: 2 x the level's width
            shapes[k] = (cmap[v.shape[0]], *v.shape[1:4], 2 * cmap[v.shape[0]])
        elif v.ndim == 5:
            shapes[k] = (cmap[v.shape[0]], *v.shape[1:4], cmap.get(v.shape[4], v.shape[4]))  # (the input conv keeps its 3 channels)
        else:
            shapes[k] = () if v.ndim == 0 else (cmap[v.shape[0]],)
    return random_state_dict({k: np.zeros(sh, np.float32) for k, sh in shapes.items()}, seed=4)


@pytest.mark.parametrize("planes,hidden,n_classes", [((12, 24, 40, 72), (10, 6), 3), ((16, 32, 64, 128), (16, 8), 2),
                                                      ((6, 10, 18, 34), (5, 3), 2)])  # (planes that are no multiples of 4: advisor, round 5)
def test_unet_of_another_architecture(backend, planes, hidden, n_classes):
    """`unet_planes`, `*_fc_planes` and the number of classes are model-config keywords (conf/training.yaml:123-127): widths no
    kernel is instantiated for run through the generic convolution, heads of other shapes as pointwise convolutions -- every block
    output and the head outputs against the oracle, and the inference tail (exp(radius) * direction, first-max class)."""
    vx = _small_batch(n=2500 if backend.type == "cpu" else 4000, seed=11)
    w = _other_architecture(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), planes, hidden, n_classes)
    oracle = uo.OracleNet(w, dtype=torch.float64)
    ref = oracle.forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w, device=backend)
    if backend.type == "cpu":
        net.use_mfma = False  # (the matrix-core kernels cost a fiber rendezvous per instruction on the CPU build and have their own tests)
    assert net.planes == list(planes) and net.generic_heads
    sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
    net.trace = {}
    out = net.forward(sp)
    for name, got in net.trace.items():
        want = oracle.trace[name].numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max(), err_msg=name)
    assert out["class_l"].shape[1] == n_classes
    for k in out:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4 * np.abs(ref[k]).max())
    net.trace = None
    r, d, c, mv, cls = net.forward_fused_tail(sp)
    mv_ref, cls_ref = uo.inference_tail(r.cpu().numpy(), d.cpu().numpy(), c.cpu().numpy())
    np.testing.assert_allclose(mv.cpu().numpy(), mv_ref, rtol=2e-6, atol=1e-30)
    np.testing.assert_array_equal(cls.cpu().numpy(), cls_ref)


def test_unet_half_precision_storage_mode(backend):
    """BASELINE.json configs[4] (an extension: the reference's inference is float32): levels with >= 16 channels keep
    features and weights in IEEE half, f16 matrix-core kernel with float32 accumulation.  Checked against the float64
    oracle that rounds weights and activations at the same places; tolerance = a few half-precision ulps of the
    tensor's scale (one rounding flip per layer can propagate), written here: 1e-3 of max|ref| on the UNet features,
    5e-3 on the head outputs (the direction head is normalised: small vectors amplify the deviation)."""
    vx = _small_batch(n=5000, seed=9)
    w = random_state_dict(uo.load_weights(WEIGHTS / "peach-forest-65.npz"), seed=2)
    oracle16 = uo.OracleNet(w, dtype=torch.float64, fp16=True)
    ref16 = oracle16.forward(vx["feats"][:, :3], vx["coords"])
    tail16 = oracle16.trace["tail0"].numpy()
    oracle32 = uo.OracleNet(w, dtype=torch.float64)
    oracle32.forward(vx["feats"][:, :3], vx["coords"])
    tail32 = oracle32.trace["tail0"].numpy()
    net = Smart_Tree(w, device=backend, fp16=True)
    sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
    feats = net.features(sp)
    assert feats.dtype == torch.float32  # level 0 stays float32
    feats = feats.cpu().numpy()
    assert (tail16 > 0).mean() > 0.2, "test input does not exercise the network"
    scale = np.abs(tail16).max()
    err16 = np.abs(feats - tail16).max() / scale
    err32 = np.abs(feats - tail32).max() / scale
    assert err16 <= 1e-3, err16
    assert err32 > err16, (err16, err32)  # the half-precision restatement is closer than the float32 network
    out = net.forward(sp)
    for k in out:
        np.testing.assert_allclose(out[k].cpu().numpy(), ref16[k], rtol=0, atol=5e-3 * np.abs(ref16[k]).max())  # heads: F.normalize amplifies


def test_spatial_order_is_invisible(backend):
    """Smart_Tree runs the network on Morton-ordered rows (st_spatial_order) and scatters the result back: every output must
    equal, bit for bit, the one computed in the input's own order -- in float32 and in half-precision storage mode."""
    vx = _small_batch(n=5000, seed=9)
    coords = torch.from_numpy(vx["coords"]).to(backend)
    order = ops.spatial_order(coords).cpu().numpy().astype(np.int64)
    assert sorted(order.tolist()) == list(range(len(order)))
    c = vx["coords"][order].astype(np.int64)
    assert (np.diff(c[:, 0]) >= 0).all()  # blocks stay together ...
    same_block = np.diff(c[:, 0]) == 0
    step = np.abs(np.diff(c[:, 1:], axis=0)).sum(1)[same_block]
    assert np.median(step) <= 3  # ... and consecutive rows are spatial neighbours (input order: far apart)
    for fp16 in ((False,) if backend.type == "cpu" else (False, True)):  # (half-precision mode: on the GPU only, for time)
        w = random_state_dict(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), seed=4)
        sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
        net = Smart_Tree(w, device=backend, fp16=fp16)
        net.use_mfma = backend.type != "cpu"  # (sanitizer build: the vector kernels -- the property is about row order, not the kernel)
        assert net.spatial_order
        a = net.forward(sp)
        net.spatial_order = False
        b = net.forward(sp)
        for k in a:
            assert torch.equal(a[k], b[k]), (k, fp16)


# ---- rulebooks from occupancy bricks (csrc/brick.hip): same sets, same pairs, same network outputs as the hash-table builders ----
def _hint(coords):
    return int(coords[:, 0].max()) + 1, int(coords[:, 1:].max()) + 1


def _row_map(mine, ref):
    """perm[i] = row of mine[i] among the oracle's rows (the brick path numbers the rows of a level in brick order)."""
    key = lambda c: ((c[:, 0].astype(np.int64) * 4096 + c[:, 1]) * 4096 + c[:, 2]) * 4096 + c[:, 3]
    km, kr = key(mine), key(ref)
    assert len(np.unique(km)) == len(km) == len(kr)
    order = np.argsort(kr)
    pos = np.searchsorted(kr[order], km)
    assert np.array_equal(kr[order][pos], km), "different voxel sets"
    return order[pos]


def _renumber(table, perm_out, perm_in, n_ref_out):
    """a brick-order table [27, n] in the oracle's numbering"""
    out = np.full((27, n_ref_out), -2, np.int64)
    t = np.where(table >= 0, perm_in[np.clip(table, 0, None)], -1)
    out[:, perm_out] = t
    return out


@pytest.mark.parametrize("case", ["tree", "isolated-odd", "batch"])
def test_brick_pyramid_matches_the_oracle(backend, case):
    """Every level's active set, submanifold table, strided and inverse tables equal the oracle's after renumbering the rows
    (brick order instead of first appearance); level 0's order0 is the permutation that produces its coordinates."""
    blk_seg, n_seg = None, 1
    if case == "tree":
        coords = _small_batch()["coords"]
    elif case == "isolated-odd":  # every voxel reaches 8 outputs: the first capacity guess overflows, the call retries
        g = np.arange(7, dtype=np.int32) * 10 + 5
        z, y, x = np.meshgrid(g, g, g, indexing="ij")
        coords = np.stack([np.zeros(z.size, np.int32), z.ravel(), y.ravel(), x.ravel()], axis=1).astype(np.int32)
        coords = np.repeat(coords, 1, axis=0)
    else:  # two clouds in one batch: each cloud's coarse sets are clipped to ITS extent
        a, b = _small_batch(seed=5)["coords"], _small_batch(n=3000, seed=7)["coords"]
        b = b.copy()
        b[:, 0] += a[:, 0].max() + 1
        coords = np.concatenate([a, b]).astype(np.int32)
        seg = np.concatenate([np.zeros(a[:, 0].max() + 1, np.int32), np.ones(b[:, 0].max() + 1 - (a[:, 0].max() + 1), np.int32)])
        blk_seg, n_seg = torch.from_numpy(seg).to(backend), 2
    rng = np.random.RandomState(0)
    coords = coords[rng.permutation(len(coords))]  # any input order
    nb, bound = _hint(coords)
    got = ops.brick_pyramid(torch.from_numpy(coords).to(backend), 3, nb, bound, blk_seg, n_seg)
    assert got is not None
    pyr, order0 = got
    order0 = order0.cpu().numpy()
    assert sorted(order0.tolist()) == list(range(len(coords)))
    np.testing.assert_array_equal(pyr.coords[0].cpu().numpy(), coords[order0])
    if n_seg == 1:
        ref_levels = [coords]
        for _ in range(3):
            ref_levels.append(uo.strided_out_coords(ref_levels[-1]))
    else:  # the oracle clips a cloud's outputs to that cloud's own extent: level by level per cloud, then concatenated
        seg_of = seg[coords[:, 0]]
        per = [[coords[seg_of == s]] for s in range(2)]
        for s in range(2):
            for _ in range(3):
                per[s].append(uo.strided_out_coords(per[s][-1]))
        ref_levels = [np.concatenate([per[0][l], per[1][l]]) for l in range(4)]
    perms = []
    for level in range(4):
        mine = pyr.coords[level].cpu().numpy()
        ref = ref_levels[level]
        perms.append(_row_map(mine, ref))
        ref_subm = uo.subm_rulebook(ref)
        np.testing.assert_array_equal(_renumber(pyr.subm[level].cpu().numpy(), perms[level], perms[level], len(ref)), ref_subm)
    for level in range(3):
        fine, coarse = ref_levels[level], ref_levels[level + 1]
        if n_seg == 1:
            ref_down, ref_up = uo.down_rulebook(coarse, fine), uo.up_rulebook(fine, coarse)
        else:  # per cloud, rows shifted
            nf0, nc0 = len(per[0][level]), len(per[0][level + 1])
            d0, u0 = uo.down_rulebook(per[0][level + 1], per[0][level]), uo.up_rulebook(per[0][level], per[0][level + 1])
            d1, u1 = uo.down_rulebook(per[1][level + 1], per[1][level]), uo.up_rulebook(per[1][level], per[1][level + 1])
            ref_down = np.concatenate([d0, np.where(d1 >= 0, d1 + nf0, -1)], axis=1)
            ref_up = np.concatenate([u0, np.where(u1 >= 0, u1 + nc0, -1)], axis=1)
        np.testing.assert_array_equal(_renumber(pyr.down[level].cpu().numpy(), perms[level + 1], perms[level], len(coarse)), ref_down)
        np.testing.assert_array_equal(_renumber(pyr.up[level].cpu().numpy(), perms[level], perms[level + 1], len(fine)), ref_up)
        tagged = pyr.up_order[level].cpu().numpy()
        rows = tagged & 0x0FFFFFFF
        assert sorted(rows.tolist()) == list(range(len(fine)))
        c = pyr.coords[level].cpu().numpy()[rows]
        cls = ((c[:, 1] & 1) << 2) | ((c[:, 2] & 1) << 1) | (c[:, 3] & 1)
        assert (np.diff(cls) >= 0).all()
        np.testing.assert_array_equal((tagged >> 28) & 0xF, 8 + cls)


def test_brick_path_leaves_the_network_outputs_unchanged(backend):
    """The network on brick rulebooks (rows in brick order, coarse rows numbered differently) == the network on hash-table
    rulebooks, bit for bit: every output row is computed on its own, in k order, from the rows its table names."""
    vx = _small_batch(n=5000, seed=9)
    w = random_state_dict(uo.load_weights(WEIGHTS / "noble-elevator-58.npz"), seed=4)
    net = Smart_Tree(w, device=backend)
    net.use_mfma = backend.type != "cpu"  # (sanitizer build: the vector kernels -- the property is about the tables, not the kernel)
    plain = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend)
    hinted = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend, brick_hint=_hint(vx["coords"]))
    a = net.forward(hinted)
    b = net.forward(plain)
    wrong = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), backend, brick_hint=(1, 8))
    c = net.forward(wrong)  # a hint the input violates is noticed on the device: the hash-table path takes over
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k


# ---- split-bf16 rule-GEMM (csrc/sparse_conv.hip k_sparse_conv_mfma_b3): float32 accuracy on the bf16 matrix pipe ----
@pytest.mark.parametrize("cin,cout,c0", [(16, 16, 16), (16, 32, 16), (32, 16, 16), (32, 32, 32), (32, 64, 32), (64, 32, 32), (64, 64, 64)])
@pytest.mark.parametrize("variant", [1, 2])
def test_split_bf16_conv_has_float32_accuracy(backend, cin, cout, c0, variant):
    """Every float32 operand = three bf16 pieces that sum to it exactly; six of the nine piece products are issued.  The result
    must be as close to the float64 contraction as the float32 matrix-core kernel's (rounding-order noise), with concat,
    BatchNorm affine, residual, ReLU and a row order."""
    rng = np.random.RandomState(cin * 100 + cout + variant)
    n, n_in, K = 613, 580, 27
    nbr = rng.randint(0, n_in, size=(K, n)).astype(np.int32)
    nbr[rng.rand(K, n) < 0.4] = -1
    x = (rng.randn(n_in, cin) * np.exp(rng.uniform(-6, 6, size=(n_in, 1)))).astype(np.float32)  # rows of very different scale
    w = (rng.randn(K, cin, cout) * 0.05).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.randn(cout).astype(np.float32)
    res = rng.randn(n, cout).astype(np.float32)
    order = rng.permutation(n).astype(np.int32)
    ref = np.zeros((n, cout))
    for k in range(K):
        hit = nbr[k] >= 0
        ref[hit] += x[nbr[k][hit]].astype(np.float64) @ w[k].astype(np.float64)
    ref = np.maximum(ref * scale.astype(np.float64) + shift.astype(np.float64) + res.astype(np.float64), 0.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(backend)
    x0, x1 = (t(x[:, :c0]), t(x[:, c0:])) if c0 < cin else (t(x), None)
    args = dict(x1=x1, scale=t(scale), shift=t(shift), residual=t(res), relu=True, row_order=t(order))
    ops.B3_VARIANT = variant
    try:
        got = ops.sparse_conv(x0, t(w), t(nbr), n, wq=ops.b3_weight(t(w)), **args).cpu().numpy().astype(np.float64)
    finally:
        ops.B3_VARIANT = 0
    f32 = ops.sparse_conv(x0, t(w), t(nbr), n, wp=ops.mfma_weight(t(w)) if ops.mfma_eligible(cin, cout, c0) else None, **args)
    f32 = f32.cpu().numpy().astype(np.float64)
    # per row: rows differ in scale by e^12, so the error is measured against each row's own magnitude
    mag = np.abs(ref).max(axis=1, keepdims=True) + np.abs(shift).max() + 1e-30
    e_b3, e_f32 = (np.abs(got - ref) / mag).max(), (np.abs(f32 - ref) / mag).max()
    assert e_b3 < 1e-5 and e_b3 < 4 * e_f32 + 2e-7, (e_b3, e_f32)
