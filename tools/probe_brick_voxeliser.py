"""GPU box: the sort / brick voxeliser's front end (tools/microbench/brick_voxeliser.hip) against the library's hash insert on the same
1M-point cloud, whole-cloud mode (one membership per point), kernel by kernel (HIP events, 30 repetitions).
    python tools/probe_brick_voxeliser.py [n_points] [voxel]"""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from smart_tree_amd import _lib  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.dataset.dataset import voxelize_cloud  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
v = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
c = sample_tree_cloud(n, seed=0)
cloud = pipe.preprocessing(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
xyz = cloud.xyz.contiguous()
L = ctypes.CDLL(str(ROOT / "tools" / "microbench" / "libbrick_voxeliser.so"))
P = ctypes.c_void_p
L.bv_count.argtypes = [P, ctypes.c_int64, P, ctypes.c_float, P, P, P, P]
L.bv_scatter.argtypes = [P, ctypes.c_int64, P, ctypes.c_float, P, P, P, P, P]
L.bv_bricks.argtypes = [P, ctypes.c_int, P, P, P, P, P, P, P]
lo = xyz.min(0).values.cpu()
hi = xyz.max(0).values.cpu()
lo_arr = (ctypes.c_float * 3)(*lo.tolist())
nb = [int(((hi[a] - lo[a]) / v).floor().item()) // 8 + 1 for a in range(3)]
nb_arr = (ctypes.c_int * 3)(*nb)
NB = nb[0] * nb[1] * nb[2]
stream = _lib.stream(dev)
cnt = torch.zeros(NB + 1, dtype=torch.int32, device=dev)
rank = torch.empty(n, dtype=torch.int32, device=dev)
sorted_ = torch.empty((n, 2), dtype=torch.int32, device=dev)
n_vox = torch.zeros(1, dtype=torch.int32, device=dev)
out_b, out_c, out_r = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def count():
    cnt.zero_()
    L.bv_count(xyz.data_ptr(), n, lo_arr, v, nb_arr, cnt.data_ptr(), rank.data_ptr(), stream)


count()
base = torch.zeros(NB + 1, dtype=torch.int32, device=dev)


def scan():
    torch.cumsum(cnt[:NB], 0, out=base[1:])


scan()
bricks = (cnt[:NB] > 0).nonzero().view(-1)


def scatter():
    L.bv_scatter(xyz.data_ptr(), n, lo_arr, v, nb_arr, base.data_ptr(), rank.data_ptr(), sorted_.data_ptr(), stream)


def brick_pass():
    n_vox.zero_()
    L.bv_bricks(bricks.data_ptr(), int(bricks.numel()), base.data_ptr(), sorted_.data_ptr(), n_vox.data_ptr(), out_b.data_ptr(),
                out_c.data_ptr(), out_r.data_ptr(), stream)


t = {"count (memset + returning atomic per point)": timed(count), "scan of the brick table (torch.cumsum)": timed(scan),
     "scatter": timed(scatter), "brick pass (LDS first-point-wins + emit)": timed(brick_pass)}
scatter()
brick_pass()
vb = voxelize_cloud(xyz, cloud.rgb, v)
t_lib = timed(lambda: voxelize_cloud(xyz, cloud.rgb, v), reps=20)
print(f"{n} points, voxel {v}: brick table {nb} = {NB} entries, {bricks.numel()} occupied bricks, {int(n_vox.item())} voxels "
      f"(library, whole-cloud mode: {vb.coords.shape[0]}); representatives agree: "
      f"{bool(torch.equal(torch.sort(out_r[:int(n_vox.item())].long()).values, torch.sort(vb.point_index).values))}")
for k, us in t.items():
    print(f"  {us:8.1f} us  {k}")
print(f"  {sum(t.values()):8.1f} us  sum of the sort / brick front end (the two ordering sorts and the gather come on top, as in the library)")
print(f"  {t_lib:8.1f} us  library st_voxelize_cloud_seg, the WHOLE stage incl. ordering sorts, gather and its host read-back")
