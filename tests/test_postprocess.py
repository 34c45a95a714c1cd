"""st_post_process (prune / repair / smooth on the device, csrc/postprocess.hip) against oracle/pipeline_oracle.py on synthetic
branch tables of every size class of the kernel: a lane per branch (<= 1024 branches in a tree), the loop over branches in LDS
(<= 8192), the tables in global memory (more), and several trees in one call.  Reference: data_types/tree.py:73-134,164-176."""
import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as po
from smart_tree_amd.skeleton.skeletonize import DeviceSkeleton


def _random_tree(rng, nb, first_len=40):
    """Branch hierarchy with parent id < child id; a child starts near a random vertex of its parent."""
    branches = []
    for b in range(nb):
        n = int(rng.integers(2, 14)) if b else first_len
        if b == 0:
            parent, origin = -1, np.zeros(3, np.float32)
        else:
            parent = int(rng.integers(0, b)) if rng.random() < 0.9 else int(rng.integers(max(0, b - 4), b))  # deep chains too
            pv = branches[parent][1]
            origin = pv[int(rng.integers(0, len(pv)))] + rng.normal(0, 0.01, 3).astype(np.float32)
        step = rng.normal(0, 1, 3)
        step = (0.03 * step / np.linalg.norm(step)).astype(np.float32)
        xyz = (origin + np.arange(1, n + 1, dtype=np.float32)[:, None] * step + rng.normal(0, 0.004, (n, 3))).astype(np.float32)
        # a few very short / thin branches so that prune removes some (and their whole sub-hierarchy)
        scale = 0.05 if rng.random() < 0.05 else 1.0
        xyz = (origin + (xyz - origin) * np.float32(scale)).astype(np.float32)
        rad = rng.uniform(0.004 if rng.random() < 0.05 else 0.012, 0.05, n).astype(np.float32)
        branches.append((parent, xyz, rad))
    return branches


def _device_skeleton(trees, device):
    tree_off, parent, start, length, xyz, rad = [0], [], [], [], [], []
    slot = 0
    for branches in trees:
        for par, bx, br in branches:
            parent.append(par)
            start.append(slot)
            length.append(len(bx))
            xyz.append(np.concatenate([bx[:1], bx]))  # slot start[k]: reserved for the connection point (pre-filled, tree.py:92)
            rad.append(np.concatenate([br[:1], br]))
            slot += len(bx) + 1
        tree_off.append(len(parent))
    i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=device)
    return DeviceSkeleton(i32(tree_off), i32(parent), i32(start), i32(length),
                          torch.from_numpy(np.concatenate(xyz)).to(device), torch.from_numpy(np.concatenate(rad)).to(device))


@pytest.mark.parametrize("sizes", [(300,), (1500,), (9000,), (700, 1, 40, 1100, 3)])
def test_post_process_matches_the_oracle(backend, sizes):
    rng = np.random.default_rng(len(sizes) * 1000 + sizes[0])
    trees = [_random_tree(rng, nb) for nb in sizes]
    sk = _device_skeleton(trees, backend)
    sk.prune(min_radius=0.01, min_length=0.02)
    sk.repair()
    sk.smooth(kernel_size=5)
    ref = [po.OTree(t, {b: po.OBranch(b, par, bx.copy(), br.reshape(-1, 1).copy()) for b, (par, bx, br) in enumerate(branches)})
           for t, branches in enumerate(trees)]
    po.post_process(ref, True, 0.01, 0.02, True, True, 5)
    got = sk.skeletons
    assert len(got) == len(ref)
    pruned = 0
    for g, r in zip(got, ref):
        assert list(g.branches.keys()) == list(r.branches.keys())
        for k, rb in r.branches.items():
            gb = g.branches[k]
            assert gb.parent_id == rb.parent_id
            np.testing.assert_array_equal(gb.xyz.numpy(), rb.xyz)
            np.testing.assert_array_equal(gb.radii.numpy(), rb.radii)
    pruned = sizes[0] - len(ref[0].branches)
    assert 0 < pruned < sizes[0]  # the case exercises the keep chain (only skeleton 0 is pruned: tree.py:164-168)
    assert all(len(r.branches) == nb for r, nb in zip(ref[1:], sizes[1:]))
