"""Helper workgroups of the branch selection's long-path claims (csrc/skeleton.hip, k_sk_select): the inter-workgroup protocol and
every one of its fall-backs, driven against the SHIPPED library through the per-call tuning codes 16-18 (time-outs in microseconds).

The reference has no counterpart (sample_tree, skeleton/path.py:49-140, is a host loop); what is pinned here is that the result --
branch ids, parents, vertex lists -- never depends on whether helpers took part, left half-way or were never waited for.
The CPU emulator reports one compute unit, so the library launches no helpers there: GPU only."""
import numpy as np
import pytest
import torch

from oracle import skeleton_oracle as so
from oracle import voxel_oracle as vo
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.skeleton import skeletonize, tuning
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def branch_cloud():
    """configs[1]'s tree with its ground-truth medial vectors: one component of ~68k vertices whose trunk and limbs are claimed
    in long mode (the jobs helpers take shares of)."""
    c = sample_tree_cloud(1_000_000, seed=0)
    vx = vo.voxelize_cloud(vo.centre_cloud(c["xyz"]), c["rgb"], 0.02)
    m = vx["mask"]
    return vx["feats"][m, :3], c["medial_vector"][vx["point"][m]]


def _signature(out):
    sig = []
    for tree in out.skeletons:
        for b in tree.branches.values():
            sig.append((tree._id, b._id, b.parent_id, b.xyz.numpy().tobytes(), b.radii.numpy().tobytes()))
    return sig


def _run(pts, mv, knobs):
    dev = torch.device("cuda:0")
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
    skeletonize.reset_helper_stats()
    with tuning.override(knobs):
        out = sk.forward(Cloud(xyz=torch.from_numpy(pts).to(dev), medial_vector=torch.from_numpy(mv).to(dev)))
        sig = _signature(out)
    return sig, skeletonize.helper_stats()


def test_product_settings_use_helpers_and_lose_none(branch_cloud):
    pts, mv = branch_cloud
    sig, st = _run(pts, mv, {})
    assert st["with_helpers"] == 1 and st["lost"] == 0, st
    ref = so.skeletonize(pts, mv)
    kept = np.nonzero(ref.keep_mask)[0]
    medial = (pts + mv)[kept]
    want = []
    for t, rc in enumerate(ref.components):
        for b in rc.branches:
            want.append((t, b.branch_id, b.parent_id, medial[rc.vertex_ids[b.verts]].astype(np.float32).tobytes()))
    assert [(t, i, p, x) for t, i, p, x, _ in sig] == want


@pytest.mark.parametrize("knobs,expect_lost", [
    ({12: 256}, False),                                       # no helpers at all: the component's workgroup claims alone
    ({tuning.HELP_LIFETIME_US: 0}, True),                     # helpers expire at once
    ({tuning.HELP_LIFETIME_US: 1000}, True),                  # ... after a millisecond: in the middle of the jobs
    ({tuning.HELP_TIMEOUT_US: 0}, True),                      # the component's workgroup never waits for its (live) helpers
    ({tuning.HELP_ANNOUNCE_US: 0}, None),                     # helpers give up unless their component is resident already
    ({12: (1 + 4) << 8}, False),                              # four helpers, forced
    ({12: (1 + 4000) << 8}, False),                           # more than the device may lend: capped
], ids=["none", "expire-at-once", "expire-1ms", "never-wait", "announce-0", "forced-4", "capped"])
def test_fallbacks_keep_the_result(branch_cloud, knobs, expect_lost):
    pts, mv = branch_cloud
    base, st0 = _run(pts, mv, {})
    assert st0["lost"] == 0
    got, st = _run(pts, mv, knobs)
    assert got == base
    if expect_lost is not None:
        assert (st["lost"] > 0) == expect_lost, st
    if knobs.get(12) == 256:
        assert st["with_helpers"] == 0


def test_two_calls_in_flight_share_the_helper_budget(branch_cloud):
    """Two helper-enabled calls on different streams (bench.py keeps batches in flight): the process-wide lease keeps the sum of
    their waiting workgroups under the device's cap, results are the serial ones, nothing is lost."""
    import threading

    pts, mv = branch_cloud
    base, _ = _run(pts, mv, {})
    dev = torch.device("cuda:0")
    skeletonize.reset_helper_stats()
    got, errors = {}, []

    def worker(w):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
                for r in range(3):
                    with tuning.override({12: (1 + 100) << 8}):  # each asks for 100 of the 160
                        out = sk.forward(Cloud(xyz=torch.from_numpy(pts).to(dev), medial_vector=torch.from_numpy(mv).to(dev)))
                    got[(w, r)] = _signature(out)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(sig == base for sig in got.values())
    st = skeletonize.helper_stats()
    assert st["calls"] == 6 and st["lost"] == 0, st
