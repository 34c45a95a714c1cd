"""Developer aid: compare the GPU skeleton stage with the oracle on the bench cloud, branch by branch."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import pipeline_oracle as po, unet_oracle as uo, skeleton_oracle as so
from smart_tree_amd.synthetic import sample_tree_cloud
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.data_types.cloud import Cloud
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device('cuda:0')
c = sample_tree_cloud(n, 0)
w = uo.load_weights('smart_tree_amd/model/weights/noble-elevator-58.npz')
lc = po.labelled_cloud(c['xyz'], c['rgb'], w, 0.02)
m = np.isin(lc['class_l'].reshape(-1), [0])
pts, mv = lc['xyz'][m].astype(np.float32), lc['medial_vector'][m].astype(np.float32)
ref = so.skeletonize(pts, mv)
print('ref comps', [len(c.vertex_ids) for c in ref.components], 'branches', [len(c.branches) for c in ref.components], flush=True)
sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
kept = np.nonzero(ref.keep_mask)[0]
medial = (pts + mv)[kept]
for rep in range(3):
    out = sk.forward(Cloud(xyz=torch.from_numpy(pts).to(dev), medial_vector=torch.from_numpy(mv).to(dev)))
    bad = 0
    for tree, rc in zip(out.skeletons, ref.components):
        for b in rc.branches:
            g = tree.branches.get(b.branch_id)
            if g is None: print('missing', b.branch_id); bad += 1; continue
            if g.parent_id != b.parent_id: print('parent', b.branch_id, 'got', g.parent_id, 'ref', b.parent_id); bad += 1
            if g.xyz.shape[0] != len(b.verts) or not np.array_equal(g.xyz.cpu().numpy(), medial[rc.vertex_ids[b.verts]]):
                print('verts', b.branch_id, g.xyz.shape[0], len(b.verts)); bad += 1
            if bad > 12: break
    print('rep', rep, 'bad', bad, flush=True)
