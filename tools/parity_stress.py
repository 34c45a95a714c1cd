"""Randomised parity sweep on the GPU (developer evidence, not part of the test suite): many random clouds through the HIP
path and the oracle.

    python tools/parity_stress.py [n_cases=60] [rng_seed=2024]

Per case (random size 20k-300k points, voxel 2-5 cm, tree shape, optional foliage):
  * CentreCloud + blocks + voxels: bit-exact against oracle/voxel_oracle.py (coords, order, masks, representatives);
  * skeleton stage on the voxel representatives with EXACT medial vectors: components, branch ids, parents and vertex lists
    identical to oracle/skeleton_oracle (C), post-processed geometry identical to oracle/pipeline_oracle;
  * every third case: the same clouds in one batch (Pipeline.process_clouds with the shipped checkpoint) == one at a time.
"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from oracle import pipeline_oracle as po, skeleton_oracle as so, voxel_oracle as vo
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = np.random.RandomState(SEED)
pipe = bench.build_pipeline(dev)
sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
t0 = time.time()
stats = {"cases": 0, "voxels": 0, "branches": 0, "components": 0, "batch_checks": 0}
pending = []


def signature(s):
    return [(t._id, b._id, b.parent_id, b.xyz.numpy().tobytes(), b.radii.numpy().tobytes()) for t in s.skeletons for b in t.branches.values()]


for case in range(N):
    n = int(rng.randint(20_000, 300_000))
    voxel = float(rng.choice([0.02, 0.025, 0.03, 0.04, 0.05]))
    kw = dict(seed=(1000 if SEED == 2024 else 100_000 + 7919 * SEED) + case, scale=float(rng.uniform(0.5, 1.2)), max_depth=int(rng.randint(4, 8)))
    if rng.rand() < 0.3:
        kw["foliage_fraction"] = float(rng.uniform(0.1, 0.5))
    c = sample_tree_cloud(n, **kw)
    xyz = vo.centre_cloud(c["xyz"])
    ref = vo.voxelize_cloud(xyz, c["rgb"], voxel)
    got = voxelize_blocks(torch.from_numpy(xyz).to(dev), torch.from_numpy(c["rgb"]).to(dev), voxel)
    assert np.array_equal(got.coords.cpu().numpy(), ref["coords"]) and np.array_equal(got.point_index.cpu().numpy(), ref["point"])
    assert np.array_equal(got.mask.cpu().numpy(), ref["mask"]) and np.array_equal(got.feats.cpu().numpy(), ref["feats"])
    m = ref["mask"]
    pts, mv = ref["feats"][m, :3], c["medial_vector"][ref["point"][m]]
    r = so.skeletonize(pts, mv)
    out = sk.forward(Cloud(xyz=torch.from_numpy(pts).to(dev), medial_vector=torch.from_numpy(mv).to(dev)))
    assert len(out.skeletons) == len(r.components), (case, len(out.skeletons), len(r.components))
    kept = np.nonzero(r.keep_mask)[0]
    medial = (pts + mv)[kept]
    for tree, rc in zip(out.skeletons, r.components):
        assert list(tree.branches) == [b.branch_id for b in rc.branches], case
        for b in rc.branches:
            g = tree.branches[b.branch_id]
            assert g.parent_id == b.parent_id and np.array_equal(g.xyz.numpy(), medial[rc.vertex_ids[b.verts]]), case
        stats["branches"] += len(rc.branches)
    stats["components"] += len(r.components)
    stats["voxels"] += len(ref["coords"])
    stats["cases"] += 1
    pending.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
    if len(pending) == 3:
        pipe.model_inference.voxel_size = voxel
        serial = [signature(pipe.process_cloud(cloud=p)) for p in pending]
        for one, part in zip(serial, pipe.process_clouds(pending)):
            assert signature(part) == one, case
        stats["batch_checks"] += 1
        pending = []
print(f"parity_stress: {stats} in {time.time() - t0:.0f} s -- all identical")
