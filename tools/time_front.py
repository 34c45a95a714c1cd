"""Dev script: time voxelize + UNet on a 1M-point synthetic tree on cuda:0."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from oracle import voxel_oracle as vo, unet_oracle as uo
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.model.model import Smart_Tree
from smart_tree_amd.model.sparse import sparse_from_batch
from smart_tree_amd.model import sparse_ops as ops
from smart_tree_amd.synthetic import sample_tree_cloud

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
c = sample_tree_cloud(n, 0)
xyz = torch.from_numpy(vo.centre_cloud(c["xyz"])).to(dev); rgb = torch.from_numpy(c["rgb"]).to(dev)
w = uo.load_weights(ROOT / "smart_tree_amd/model/weights/noble-elevator-58.npz")
net = Smart_Tree(w, device=dev)
def sync(): torch.cuda.synchronize()
for it in range(4):
    sync(); t0 = time.perf_counter()
    vb = voxelize_blocks(xyz, rgb, 0.02); sync(); t1 = time.perf_counter()
    sp = sparse_from_batch(vb.feats[:, :3].contiguous(), vb.coords, dev); sync(); t2 = time.perf_counter()
    pyr = ops.build_pyramid(sp.indices, 3); sync(); t3 = time.perf_counter()
    out = net.forward_fused_tail(sp); sync(); t4 = time.perf_counter()
    print(f"iter {it}: voxelize {1e3*(t1-t0):.2f} ms  wrap {1e3*(t2-t1):.2f}  pyramid(alone) {1e3*(t3-t2):.2f}  unet(incl pyramid) {1e3*(t4-t3):.2f} ms; voxels {vb.coords.shape[0]} levels {[c.shape[0] for c in pyr.coords]}")
ref = vo.voxelize_cloud(xyz.cpu().numpy(), rgb.cpu().numpy(), 0.02)
print("voxel parity at full size:", np.array_equal(ref["coords"], vb.coords.cpu().numpy()), np.array_equal(ref["point"], vb.point_index.cpu().numpy()))
