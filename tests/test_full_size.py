"""BASELINE.json full-size checks (GPU only): configs[1] (1M-point tree, 2 cm) stage by stage against the
oracle, plus size-independent properties on a configs[3]-style dense canopy (5M points, 1 cm)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from oracle import skeleton_oracle as so
from oracle import voxel_oracle as vo
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.model import sparse_ops as ops
from oracle import unet_oracle as uo
from smart_tree_amd.model.model import Smart_Tree
from smart_tree_amd.model.model_inference import ModelInference
from smart_tree_amd.model.sparse import sparse_from_batch
from smart_tree_amd.pipeline import Pipeline
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

pytestmark = pytest.mark.gpu
WEIGHTS = Path(__file__).resolve().parents[1] / "smart_tree_amd" / "model" / "weights" / "noble-elevator-58.npz"


def _rms(a):
    return float(np.sqrt(np.mean(np.square(a, dtype=np.float64)))) + 1e-30


def _record_tolerance(what, err, base, scale):
    """The measured numbers behind the relaxed fp32 bar (`max(1e-4, 4 x the fp32 oracle's own distance from fp64)`), written
    where the builder collects them (gpurun_out/ -> profiles/r03_fp32_tolerance.txt) so the relaxation is a documented figure."""
    out = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        with open(out / "fp32_tolerance.txt", "a") as f:
            f.write(f"{what}: max|hip - fp64 oracle| / rms(fp64) = {err:.3e}; max|fp32 oracle - fp64 oracle| / rms = {base:.3e}; "
                    f"rms(fp64 medial vector) = {scale:.4e}; bar = max(1e-4, 4 x base) = {max(1e-4, 4 * base):.3e}\n")
    except OSError:
        pass


def _pipeline(dev, voxel):
    mi = ModelInference("unused", WEIGHTS, voxel_size=voxel, block_size=4, buffer_size=0.4, device=dev)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
    return Pipeline(AugmentationPipeline([CentreCloud()]), mi, sk, repair_skeletons=True, smooth_skeletons=True,
                    smooth_kernel_size=11, prune_skeletons=True, min_skeleton_radius=0.01, min_skeleton_length=0.02, device=dev)


def test_config1_million_point_tree_stagewise():
    dev = torch.device("cuda:0")
    c = sample_tree_cloud(1_000_000, seed=0)
    pipe = _pipeline(dev, 0.02)
    skeleton = pipe.process_cloud(cloud=Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    # voxelisation: bit-exact against the oracle at full size
    xyz = vo.centre_cloud(c["xyz"])
    ref = vo.voxelize_cloud(xyz, c["rgb"], 0.02)
    got = voxelize_blocks(torch.from_numpy(xyz).to(dev), torch.from_numpy(c["rgb"]).to(dev), 0.02)
    np.testing.assert_array_equal(got.coords.cpu().numpy(), ref["coords"])
    np.testing.assert_array_equal(got.point_index.cpu().numpy(), ref["point"])
    lc = pipe.last_labelled_cloud
    np.testing.assert_array_equal(lc.xyz.cpu().numpy(), ref["feats"][ref["mask"], :3])
    # network outputs of the shipped checkpoint at full size vs the float64 oracle network (same bar as tests/test_unet.py:
    # 1e-4 of the output's rms, or 4x the float32 oracle's own distance from float64 -- the checkpoint's BatchNorm
    # statistics make any fp32 evaluation order noisy)
    w = uo.load_weights(WEIGHTS)
    o64 = uo.OracleNet(w, dtype=torch.float64).forward(ref["feats"][:, :3], ref["coords"])
    o32 = uo.OracleNet(w, dtype=torch.float32).forward(ref["feats"][:, :3], ref["coords"])
    mv64, cls64 = uo.inference_tail(o64["radius"], o64["direction"], o64["class_l"])
    mv32, _ = uo.inference_tail(o32["radius"].astype(np.float64), o32["direction"].astype(np.float64), o32["class_l"])
    inner = ref["mask"]
    scale = _rms(mv64[inner])
    err = np.abs(lc.medial_vector.cpu().numpy() - mv64[inner]).max() / scale
    base = np.abs(mv32[inner] - mv64[inner]).max() / scale
    _record_tolerance("test_config1_million_point_tree_stagewise (noble-elevator-58, %d voxels)" % len(inner), err, base, scale)
    assert err <= max(1e-4, 4 * base), (err, base)
    assert (lc.class_l.cpu().numpy() != cls64[inner]).mean() < 1e-3
    # skeleton + post-processing from the SAME labelled cloud: identical to the oracle
    trees = po.skeleton_from_labelled(lc.xyz.cpu().numpy(), lc.medial_vector.cpu().numpy(), lc.class_l.cpu().numpy())
    po.post_process(trees, True, 0.01, 0.02, True, True, 11)
    assert len(skeleton.skeletons) == len(trees) >= 1
    n_branches = 0
    for got_tree, rt in zip(skeleton.skeletons, trees):
        assert list(got_tree.branches) == list(rt.branches)
        for k, rb in rt.branches.items():
            gb = got_tree.branches[k]
            assert gb.parent_id == rb.parent_id
            np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
            np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)
        n_branches += len(rt.branches)
    assert n_branches > 100


def test_config1_batch_of_million_point_trees_against_the_oracle():
    """Three 1M-point trees (seeds 1-3: 320-545 branches) through ONE launch set -- the helper workgroups of the branch
    selection's long-path claims are active for a call of this size, on the GPU only -- and every cloud's skeleton equals the
    oracle's skeleton + post-processing of that cloud's labelled points, branch by branch (ids, parents, coordinates, radii)."""
    dev = torch.device("cuda:0")
    pipe = _pipeline(dev, 0.02)
    clouds = []
    for seed in (1, 2, 3):
        c = sample_tree_cloud(1_000_000, seed=seed)
        clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    parts = pipe.process_clouds(clouds)
    lc = pipe.last_labelled_cloud
    assert len(parts) == 3 and lc.n_seg == 3
    off = lc.seg_off.cpu().tolist()
    total = 0
    for b, got in enumerate(parts):
        sl = slice(off[b], off[b + 1])
        trees = po.skeleton_from_labelled(lc.xyz[sl].cpu().numpy(), lc.medial_vector[sl].cpu().numpy(), lc.class_l[sl].cpu().numpy())
        po.post_process(trees, True, 0.01, 0.02, True, True, 11)
        assert len(got.skeletons) == len(trees) >= 1
        for got_tree, rt in zip(got.skeletons, trees):
            assert list(got_tree.branches) == list(rt.branches)
            for k, rb in rt.branches.items():
                gb = got_tree.branches[k]
                assert gb.parent_id == rb.parent_id
                np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
                np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)
            total += len(rt.branches)
    assert total > 600


def test_config3_dense_canopy_properties():
    """5M points, 60 % foliage, 1 cm voxels: too big for the oracle in seconds -> invariants instead."""
    dev = torch.device("cuda:0")
    c = sample_tree_cloud(5_000_000, seed=3, foliage_fraction=0.6)
    xyz = torch.from_numpy(vo.centre_cloud(c["xyz"])).to(dev)
    vb = voxelize_blocks(xyz, None, 0.01)
    coords = vb.coords.long()
    m = coords.shape[0]
    assert m > 1_000_000
    # (1) voxels are unique per block and ordered by (block, representative point)
    key = ((coords[:, 0] * 1024 + coords[:, 1]) * 1024 + coords[:, 2]) * 1024 + coords[:, 3]
    assert torch.unique(key).numel() == m
    order_key = coords[:, 0] * (1 << 32) + vb.point_index
    assert bool((order_key[1:] > order_key[:-1]).all())
    # (2) near-idempotence: the inner representatives, voxelised again, keep (almost) all of their own voxels
    #     (block origins move with the halo members, so a few may merge -- never more than before)
    inner = int(vb.mask.sum())
    again = voxelize_blocks(vb.feats[vb.mask][:, :3].contiguous(), None, 0.01)
    assert 0.9 * inner <= int(again.mask.sum()) <= inner
    # (3) rulebook symmetry: subm pairs are mutual with mirrored offsets; strided pairs match their transpose
    pyr = ops.build_pyramid(vb.coords, 3)
    nbr = pyr.subm[0]
    # BASELINE.json configs[3]: "rulebook > 100M active pairs" = the active (in, out) pairs the 14 submanifold 3x3x3 convolutions of
    # one forward pass work through (model_blocks.py:107-156, 159-243: a head ResBlock on every level and a tail ResBlock on levels
    # 0-2, two convolutions each -> 4 convs on levels 0-2, 2 on level 3)
    convs_per_level = (4, 4, 4, 2)
    pairs = [int((pyr.subm[l] >= 0).sum()) for l in range(4)]
    assert sum(p * k for p, k in zip(pairs, convs_per_level)) > 100_000_000, pairs
    rng = torch.randint(0, m, (200_000,), device=dev)
    for k in (0, 5, 13, 20, 26):
        j = nbr[k, rng].long()
        ok = j >= 0
        back = nbr[26 - k, j[ok]].long()
        assert bool((back == rng[ok]).all())
    down, up = pyr.down[0], pyr.up[0]
    assert int((down >= 0).sum()) == int((up >= 0).sum())
    # (4) the whole pipeline runs and the skeleton is a forest: parents precede children
    pipe = _pipeline(dev, 0.01)
    sk = pipe.process_cloud(cloud=Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    for tree in sk.skeletons:
        for b in tree.branches.values():
            assert b.parent_id < b._id and b.xyz.shape[0] == b.radii.shape[0] and torch.isfinite(b.xyz).all()
    # (5) the skeleton half at FULL size against the oracle (skeleton/skeletonize.py:31-95, skeleton/path.py:49-140): the oracle's
    #     skeleton + post-processing of the GPU's own labelled cloud equals the GPU's skeleton, branch by branch
    lc = pipe.last_labelled_cloud
    trees = po.skeleton_from_labelled(lc.xyz.cpu().numpy(), lc.medial_vector.cpu().numpy(), lc.class_l.cpu().numpy())
    po.post_process(trees, True, 0.01, 0.02, True, True, 11)
    assert len(sk.skeletons) == len(trees) >= 1
    n_branches = 0
    for got_tree, rt in zip(sk.skeletons, trees):
        assert list(got_tree.branches) == list(rt.branches)
        for k, rb in rt.branches.items():
            gb = got_tree.branches[k]
            assert gb.parent_id == rb.parent_id
            np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
            np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)
        n_branches += len(rt.branches)
    assert n_branches >= 1


def test_config1_brick_rulebooks_equal_hash_rulebooks_full_size():
    """The two rulebook builders (occupancy bricks, csrc/brick.hip; hash tables, csrc/rulebook.hip) number the rows of the
    coarse levels differently; the labelled cloud of the 1M-point tree (176k voxels, 21 blocks) must come out bit-identical."""
    dev = torch.device("cuda:0")
    c = sample_tree_cloud(1_000_000, seed=0)
    cloud = AugmentationPipeline([CentreCloud()])(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    mi = ModelInference("unused", WEIGHTS, voxel_size=0.02, block_size=4, buffer_size=0.4, device=dev)
    assert mi.model.use_bricks
    a = mi.forward(cloud)
    mi.model.use_bricks = False
    b = mi.forward(cloud)
    assert len(a) == len(b) > 100_000
    assert torch.equal(a.xyz, b.xyz) and torch.equal(a.medial_vector, b.medial_vector) and torch.equal(a.class_l, b.class_l)


def test_config3_style_canopy_against_the_oracle():
    """configs[3]'s regime at a size the oracle finishes in seconds: 400k points, 60 % foliage, 1 cm voxels, and radius
    outliers injected into the network's medial vectors (the full-size cloud has a median radius of 6 cm and a maximum of
    1.0 m: the case that made every claim / search cell too coarse in round 2).  Voxels bit-exact; skeleton + post-processing
    of the SAME labelled cloud identical to the oracle, branch by branch."""
    dev = torch.device("cuda:0")
    c = sample_tree_cloud(400_000, seed=3, foliage_fraction=0.6)
    pipe = _pipeline(dev, 0.01)
    pipe.process_cloud(cloud=Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    xyz = vo.centre_cloud(c["xyz"])
    ref = vo.voxelize_cloud(xyz, c["rgb"], 0.01)
    got = voxelize_blocks(torch.from_numpy(xyz).to(dev), torch.from_numpy(c["rgb"]).to(dev), 0.01)
    np.testing.assert_array_equal(got.coords.cpu().numpy(), ref["coords"])
    np.testing.assert_array_equal(got.point_index.cpu().numpy(), ref["point"])
    lc = pipe.last_labelled_cloud
    np.testing.assert_array_equal(lc.xyz.cpu().numpy(), ref["feats"][ref["mask"], :3])
    mv = lc.medial_vector.clone()
    rng = np.random.RandomState(5)
    hit = torch.from_numpy(rng.choice(len(lc), size=max(len(lc) // 400, 8), replace=False)).to(dev)
    mv[hit] *= torch.from_numpy(rng.uniform(4.0, 16.0, (len(hit), 1)).astype(np.float32)).to(dev)  # radii up to ~1 m
    labelled = Cloud(xyz=lc.xyz, rgb=lc.rgb, medial_vector=mv, class_l=lc.class_l)
    rad = mv.norm(dim=1)
    assert float(rad.max()) > 8 * float(rad.median())
    skeleton = pipe.skeletonizer.forward(labelled.filter_by_class(pipe.branch_classes))
    pipe.post_process(skeleton)
    trees = po.skeleton_from_labelled(lc.xyz.cpu().numpy(), mv.cpu().numpy(), lc.class_l.cpu().numpy())
    po.post_process(trees, True, 0.01, 0.02, True, True, 11)
    assert len(skeleton.skeletons) == len(trees) >= 1
    n_branches = 0
    for got_tree, rt in zip(skeleton.skeletons, trees):
        assert list(got_tree.branches) == list(rt.branches)
        for k, rb in rt.branches.items():
            gb = got_tree.branches[k]
            assert gb.parent_id == rb.parent_id
            np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
            np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)
        n_branches += len(rt.branches)
    assert n_branches > 20


def test_config1_ground_truth_medial_many_components():
    """The 1M-point tree's voxel representatives with EXACT medial vectors: ~70k graph vertices in dozens of
    components of very different sizes -- all advanced in lockstep by the skeleton kernels; must equal the oracle."""
    dev = torch.device("cuda:0")
    c = sample_tree_cloud(1_000_000, seed=0)
    vx = vo.voxelize_cloud(vo.centre_cloud(c["xyz"]), c["rgb"], 0.02)
    m = vx["mask"]
    pts, mv = vx["feats"][m, :3], c["medial_vector"][vx["point"][m]]
    ref = so.skeletonize(pts, mv)
    assert len(ref.components) > 10
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
    for _ in range(2):  # twice: results must not depend on scheduling
        out = sk.forward(Cloud(xyz=torch.from_numpy(pts).to(dev), medial_vector=torch.from_numpy(mv).to(dev)))
        assert len(out.skeletons) == len(ref.components)
        kept = np.nonzero(ref.keep_mask)[0]
        medial = (pts + mv)[kept]
        for tree, rc in zip(out.skeletons, ref.components):
            assert list(tree.branches) == [b.branch_id for b in rc.branches]
            for b in rc.branches:
                g = tree.branches[b.branch_id]
                assert g.parent_id == b.parent_id
                np.testing.assert_array_equal(g.xyz.numpy(), medial[rc.vertex_ids[b.verts]])


def test_clouds_in_flight_on_separate_streams_match_serial_results():
    """bench.py keeps several clouds in flight per GPU (one host thread + HIP stream each): the library keeps all of its
    state in caller-provided workspaces, so concurrent calls must produce exactly the serial results."""
    import threading

    dev = torch.device("cuda:0")
    clouds = []
    for seed in (5, 6):
        c = sample_tree_cloud(200_000, seed=seed)
        clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))

    def signature(sk):
        out = []
        for tree in sk.skeletons:
            for b in tree.branches.values():
                out.append((b._id, b.parent_id, b.xyz.numpy().tobytes(), b.radii.numpy().tobytes()))
        return out

    serial = [signature(_pipeline(dev, 0.02).process_cloud(cloud=c)) for c in clouds]
    assert all(len(s) > 10 for s in serial)
    torch.cuda.synchronize()
    S, rounds = 3, 4
    pipes = [_pipeline(dev, 0.02) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    got, errors = {}, []

    def worker(w):
        try:
            with torch.cuda.stream(streams[w]):
                for r in range(rounds):
                    i = (w + r) % len(clouds)
                    got[(w, r)] = (i, signature(pipes[w].process_cloud(cloud=clouds[i])))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(S)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(got) == S * rounds
    for (w, r), (i, sig) in got.items():
        assert sig == serial[i], f"worker {w} round {r} differs from the serial result of cloud {i}"


# ---------------------------------------------------------------------------------------------------------------
# Full-size network parity with LIVE channels.  The shipped checkpoints saturate on the synthetic trees (the output
# is a function of each voxel's own xyz: a wrong neighbour gather at 176k voxels would change nothing a test looks
# at), so the configs[1] voxel set goes through the network with the well-conditioned random state_dict of
# tests/test_unet.py and EVERY block output is compared with the float64 oracle.
@pytest.fixture(scope="module")
def million_point_voxels():
    c = sample_tree_cloud(1_000_000, seed=0)
    vx = vo.voxelize_cloud(vo.centre_cloud(c["xyz"]), c["rgb"], 0.02)
    assert vx["coords"].shape[0] > 150_000
    return vx


LAYERS = ["input"] + [f"{k}{l}" for l in range(4) for k in ("head", "enc", "dec", "tail") if not (l == 3 and k != "head")]


def test_config1_unet_every_layer_live_weights_full_size(million_point_voxels):
    """fp32 bar (BASELINE.json north_star: 1e-4 relative): |hip - oracle64| <= 1e-4 * |oracle64| + 1e-4 * rms(layer) for
    every element of every block output (all four levels, strided + inverse convs in parity order), on both conv paths
    (f32 matrix-core kernel and the VALU kernel).  Guard: the same network on x-mirrored coordinates must differ by
    >> tolerance, i.e. the input does exercise the neighbour gathers."""
    from test_unet import random_state_dict

    vx = million_point_voxels
    dev = torch.device("cuda:0")
    w = random_state_dict(uo.load_weights(WEIGHTS), seed=1)
    oracle = uo.OracleNet(w, dtype=torch.float64)
    ref = oracle.forward(vx["feats"][:, :3], vx["coords"])
    for name in LAYERS:
        assert (oracle.trace[name].numpy() > 0).mean() > 0.2, f"{name}: the test input does not exercise the network"
    sp = sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), dev)
    for use_mfma in (True, False):
        net = Smart_Tree(w, device=dev)
        net.use_mfma = use_mfma
        net.trace = {}
        out = net.forward(sp)
        assert set(net.trace) == set(LAYERS)
        for name in LAYERS:
            r = oracle.trace[name].numpy()
            g = net.trace[name].cpu().numpy()
            assert g.shape == r.shape
            np.testing.assert_allclose(g, r, rtol=1e-4, atol=1e-4 * _rms(r), err_msg=f"{name} (mfma={use_mfma})")
        for k in out:
            g, r = out[k].cpu().numpy(), ref[k]
            if k == "direction":
                # F.normalize divides by |v|, which for a few voxels is ~100x below the head's rms: the 1e-4 bar holds for
                # all but <= 1e-4 of the elements, and those stay within 2e-3 absolute (unit vectors)
                bad = np.abs(g - r) > 1e-4 * np.abs(r) + 1e-4 * _rms(r)
                assert bad.mean() <= 1e-4 and np.abs(g - r).max() <= 2e-3, (bad.sum(), np.abs(g - r).max())
            else:
                np.testing.assert_allclose(g, r, rtol=1e-4, atol=1e-4 * _rms(r), err_msg=k)
    # guard: mirrored x coordinates -> same voxels, different neighbourhoods
    mirrored = vx["coords"].copy()
    mirrored[:, 3] = mirrored[:, 3].max() - mirrored[:, 3]
    other = uo.OracleNet(w, dtype=torch.float32)
    other.forward(vx["feats"][:, :3], mirrored)
    for name in ("head0", "tail0"):
        r = oracle.trace[name].numpy()
        assert np.abs(other.trace[name].numpy() - r).max() > 1000 * 1e-4 * _rms(r), f"{name} is blind to the gathers"
    # and the HIP net sees the same change (it is not, e.g., ignoring the rulebook)
    net = Smart_Tree(w, device=dev)
    net.trace = {}
    net.forward(sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(mirrored), dev))
    r = other.trace["tail0"].numpy()
    np.testing.assert_allclose(net.trace["tail0"].cpu().numpy(), r, rtol=1e-3, atol=1e-3 * _rms(r))


def test_config1_shipped_checkpoint_error_trace(million_point_voxels):
    """Where does the float32 distance of the SHIPPED checkpoint (noble-elevator-58) enter?  Walks every block output of the
    HIP network and of the float32 oracle network against the float64 oracle network and records, per layer,
    max|x - fp64| / rms(fp64) for both (gpurun_out/shipped_checkpoint_trace.txt -> profiles/).  The HIP network must never be
    further from float64 than 4x the float32 oracle (another float32 evaluation order of the same graph), layer by layer, and
    the first layer whose float32-order noise exceeds 1e-4 is named: that is where the bar of north_star stops being
    reachable in float32 for this checkpoint (a BatchNorm with var 6.6e-22 and |mean| 4e3)."""
    vx = million_point_voxels
    dev = torch.device("cuda:0")
    w = uo.load_weights(WEIGHTS)
    o64 = uo.OracleNet(w, dtype=torch.float64)
    r64 = o64.forward(vx["feats"][:, :3], vx["coords"])
    o32 = uo.OracleNet(w, dtype=torch.float32)
    r32 = o32.forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w, device=dev)
    net.trace = {}
    out = net.forward(sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), dev))
    lines, first_noisy = [], None
    for name in LAYERS:
        r = o64.trace[name].numpy()
        scale = _rms(r)
        e_hip = float(np.abs(net.trace[name].cpu().numpy() - r).max() / scale)
        e_f32 = float(np.abs(o32.trace[name].numpy().astype(np.float64) - r).max() / scale)
        # per channel: which output channel carries the layer's float32 noise?
        d = np.abs(o32.trace[name].numpy().astype(np.float64) - r).max(axis=0) / scale
        lines.append(f"{name:7s} rms {scale:10.4e}  hip {e_hip:9.3e}  fp32 oracle {e_f32:9.3e}  noisiest channel {int(d.argmax())} ({d.max():.3e}; "
                     f"median channel {np.median(d):.3e})")
        if first_noisy is None and e_f32 > 1e-4:
            first_noisy = name
        assert e_hip <= max(1e-4, 4 * e_f32), (name, e_hip, e_f32)
    # ... and the three heads (per-voxel 8 -> 8 -> 4 -> {1, 3, 2} with two BatchNorms; the direction is normalised): the heads of
    # the float32 oracle evaluated on the FLOAT64 network's last block output isolate what the heads alone add
    x64 = o64.trace["tail0"]
    for key, prefix in (("radius", "radius_head"), ("direction", "direction_head"), ("class_l", "class_head")):
        r = r64[key]
        scale = _rms(r)
        e_hip = float(np.abs(out[key].cpu().numpy() - r).max() / scale)
        e_f32 = float(np.abs(r32[key].astype(np.float64) - r).max() / scale)
        h = o32.head(x64.to(torch.float32), prefix)
        if key == "direction":
            h = torch.nn.functional.normalize(h)
        e_head = float(np.abs(h.numpy().astype(np.float64) - r).max() / scale)
        lines.append(f"{key:9s} rms {scale:10.4e}  hip {e_hip:9.3e}  fp32 oracle {e_f32:9.3e}  float32 head on the float64 block output {e_head:9.3e}")
        assert e_hip <= max(1e-4, 4 * e_f32), (key, e_hip, e_f32)
    # ... and the inference tail exp(radius) * direction (model_inference.py:87-88): absolute errors of the log-radius are
    # relative errors of the medial vector, and the normalised direction of a voxel whose raw direction is short is noisy
    mv64, _ = uo.inference_tail(r64["radius"], r64["direction"], r64["class_l"])
    with np.errstate(over="ignore", invalid="ignore"):
        mv_hip = np.exp(out["radius"].cpu().numpy().astype(np.float64)) * out["direction"].cpu().numpy()
        mv_f32 = np.exp(r32["radius"].astype(np.float64)) * r32["direction"]
    fin = np.isfinite(mv64).all(axis=1) & (np.abs(mv64).max(axis=1) < 1e30)
    scale = _rms(mv64[fin])
    raw64 = o64.head(x64, "direction_head").numpy()
    short = np.linalg.norm(raw64, axis=1) < 0.01 * _rms(raw64)
    for what, sel in (("all voxels with a finite float64 medial vector", fin), ("... whose raw direction is not short (>= 1 % of its rms)", fin & ~short)):
        e_hip = float(np.abs(mv_hip[sel] - mv64[sel]).max() / scale)
        e_f32 = float(np.abs(mv_f32[sel] - mv64[sel]).max() / scale)
        lines.append(f"medial vector, {what}: {int(sel.sum())} voxels, hip {e_hip:9.3e}  fp32 oracle {e_f32:9.3e}")
    lines.append(f"log-radius: rms {_rms(r64['radius']):.3e}, max |hip - fp64| {float(np.abs(out['radius'].cpu().numpy() - r64['radius']).max()):.3e} "
                 "(an absolute error of the log-radius is a relative error of the medial vector)")
    out_dir = Path(__file__).resolve().parents[1] / "gpurun_out"
    try:
        out_dir.mkdir(exist_ok=True)
        (out_dir / "shipped_checkpoint_trace.txt").write_text(
            "# tests/test_full_size.py::test_config1_shipped_checkpoint_error_trace: noble-elevator-58 on the 1M-point tree's voxels;\n"
            "# per block output: max|x - fp64 oracle| / rms(fp64) for the HIP network and for the float32 oracle network\n"
            + "\n".join(lines) + f"\nfirst layer whose float32-order noise exceeds 1e-4: {first_noisy}\n")
    except OSError:
        pass


def test_config5_half_precision_network_full_size_live_weights(million_point_voxels):
    """BASELINE.json configs[4] (extension; the reference's inference is float32): half-precision storage + f16
    matrix-core kernels on the >= 16-channel levels.  Compared with the float64 oracle that rounds weights and
    activations to half at the same places.  Tolerance per element of every block output: 4e-3 * |ref| + 1.5e-2 * rms(layer)
    (one half ulp is 4.9e-4 of a value; a rounding flip -- the HIP path accumulates in float32, the restatement in float64 --
    propagates through <= 14 layers), error rms <= 1.5e-3 of the layer's rms, and the half-precision network must be closer
    to that restatement than to the float64 network without rounding (the test can tell the two apart)."""
    from test_unet import random_state_dict

    vx = million_point_voxels
    dev = torch.device("cuda:0")
    peach = WEIGHTS.parent / "peach-forest-65.npz"
    w = random_state_dict(uo.load_weights(peach), seed=5)  # a draw whose three heads are alive on > 98 % of the voxels
    o16 = uo.OracleNet(w, dtype=torch.float64, fp16=True)
    ref16 = o16.forward(vx["feats"][:, :3], vx["coords"])
    o32 = uo.OracleNet(w, dtype=torch.float64)
    o32.forward(vx["feats"][:, :3], vx["coords"])
    net = Smart_Tree(w, device=dev, fp16=True)
    net.trace = {}
    out = net.forward(sparse_from_batch(torch.from_numpy(vx["feats"][:, :3]), torch.from_numpy(vx["coords"]), dev))
    for name in LAYERS:
        r = o16.trace[name].numpy()
        g = net.trace[name].float().cpu().numpy()
        assert (r > 0).mean() > 0.2 and g.shape == r.shape
        assert net.trace[name].dtype == (torch.float16 if r.shape[1] % 16 == 0 else torch.float32), name
        # measured on MI355X: worst element 8e-3 of the layer's rms (dec0), error rms 3e-4 of the layer's rms
        np.testing.assert_allclose(g, r, rtol=4e-3, atol=1.5e-2 * _rms(r), err_msg=name)
        assert _rms(g - r) <= 1.5e-3 * _rms(r), (name, _rms(g - r) / _rms(r))
    g, r16, r32 = net.trace["tail0"].cpu().numpy(), o16.trace["tail0"].numpy(), o32.trace["tail0"].numpy()
    assert _rms(g - r16) < 0.8 * _rms(g - r32), (_rms(g - r16), _rms(g - r32))  # measured 2.3e-4 against 3.5e-4
    for k in ("radius", "class_l"):
        err = np.abs(out[k].cpu().numpy() - ref16[k]).max()
        assert err <= 5e-3 * np.abs(ref16[k]).max(), f"{k}: {err:.2e}"
    # direction = F.normalize(v): the error of v is amplified by 1 / |v|, and a voxel whose head is dead (v == 0 exactly in the
    # restatement -> direction 0) becomes a unit vector under any perturbation: bar = 5e-3 x (typical |v| / |v|), rows with
    # |v| below a thousandth of the typical length are not comparable
    v = o16.head(o16.trace["tail0"], "direction_head").numpy()
    length = np.linalg.norm(v, axis=1)
    typical = _rms(length)
    rows = length > 1e-3 * typical
    assert rows.mean() > 0.5, "the direction head is dead on this input"
    err = np.abs(out["direction"].cpu().numpy() - ref16["direction"])[rows].max(axis=1)
    assert (err <= 5e-3 * np.maximum(1.0, typical / length[rows])).all(), float((err / np.maximum(1.0, typical / length[rows])).max())


def test_config5_half_precision_pipeline_with_the_shipped_checkpoint():
    """The peach-forest-65 checkpoint end to end in half-precision storage mode.  Its outputs are nearly constant on
    synthetic trees (radius ~ 1.9 mm, one class), so agreement with the float32 network is asserted relative to each
    output's own spread where there is one, and the degenerate case is asserted as such instead of passing vacuously."""
    dev = torch.device("cuda:0")
    peach = WEIGHTS.parent / "peach-forest-65.npz"
    c = sample_tree_cloud(1_000_000, seed=0)
    cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
    cloud = AugmentationPipeline([CentreCloud()])(cloud)
    out = {}
    for fp16 in (False, True):
        mi = ModelInference("unused", peach, voxel_size=0.02, block_size=4, buffer_size=0.4, device=dev, fp16=fp16)
        assert mi.model.fp16 == fp16
        out[fp16] = mi.forward(cloud, return_masked=False)
    a, b = out[False], out[True]
    assert len(a) == len(b) > 100_000
    assert torch.isfinite(b.medial_vector).all()
    mv_a, mv_b = a.medial_vector.cpu().numpy(), b.medial_vector.cpu().numpy()
    spread = float(np.std(np.linalg.norm(mv_a, axis=1)))
    scale = float(np.abs(mv_a).max())
    # half storage: <= 1 % of the medial vectors' own scale (measured 2e-3); if the output has no spread at all the test
    # says so (the class head of this checkpoint is constant on the synthetic trees)
    assert np.abs(mv_a - mv_b).max() <= 1e-2 * scale, (np.abs(mv_a - mv_b).max(), scale, spread)
    agree = float((a.class_l == b.class_l).float().mean())
    n_classes = int(torch.unique(a.class_l).numel())
    assert agree > 0.999, (agree, n_classes)
