"""Dev probe (verdict r4 item 6): the strided convolutions with their output rows launched in the order of their live-offset
mask.  Prints, per level: distinct masks, mean live offsets per row, mean union of live offsets per 16- / 32-row tile in table
order and in mask order, and the conv's time with / without the row order (split-bf16 kernel, two row tiles per wavefront; the
VALU kernel at level 0).

    python tools/probe_down_mask_order.py [clouds=20]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.model import sparse_ops as ops

dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
clouds = []
for b in range(B):
    c = sample_tree_cloud(1_000_000, seed=b)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
cloud = pipe.preprocessing(Cloud.collate(clouds) if B > 1 else clouds[0])
vb = voxelize_blocks(cloud.xyz, cloud.rgb, 0.02, seg_off=cloud.seg_off)
pyr = ops.brick_pyramid(vb.coords, 3, vb.block_centres.shape[0], int(round(4.8 / 0.02)) + 2, vb.blk_seg, vb.n_seg)[0]
N = [x.shape[0] for x in pyr.coords]
print("levels", N)

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

def tile_union(mask, rows):
    n = mask.shape[0] // rows * rows
    m = mask[:n].view(-1, rows)
    u = m[:, 0].clone()
    for j in range(1, rows): u |= m[:, j]
    pc = torch.zeros_like(u)
    for k in range(27): pc += (u >> k) & 1
    return pc.float().mean().item()

planes = [8, 16, 32, 64]
for lvl in range(3):
    tbl = pyr.down[lvl]  # [27, stride] int32, -1 = no pair
    nout = N[lvl + 1]
    t = tbl[:, :nout]
    mask = torch.zeros(nout, dtype=torch.int64, device=dev)
    for k in range(27): mask |= (t[k] >= 0).long() << k
    pc = torch.zeros_like(mask)
    for k in range(27): pc += (mask >> k) & 1
    order = torch.argsort(mask, stable=True)
    t_sort = timeit(lambda: torch.argsort(mask, stable=True), 5)
    print(f"L{lvl} down {planes[lvl]}->{2 * planes[lvl]}: rows {nout}, distinct masks {torch.unique(mask).numel()}, live per row {pc.float().mean().item():.2f}; "
          f"live per 16-row tile {tile_union(mask, 16):.1f} -> {tile_union(mask[order], 16):.1f} in mask order, per 32 rows {tile_union(mask, 32):.1f} -> {tile_union(mask[order], 32):.1f}; torch argsort {t_sort:.0f} us")
    cin, cout = planes[lvl], 2 * planes[lvl]
    x = torch.randn(N[lvl], cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    ro = order.to(torch.int32)
    kw = {}
    if ops.b3_eligible(cin, cout, cin):
        kw = {"wq": ops.b3_weight(w)}
        ops.B3_VARIANT = 2
    ya = ops.sparse_conv(x, w, tbl, nout, **kw)
    yb = ops.sparse_conv(x, w, tbl, nout, row_order=ro, **kw)
    t0 = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, **kw))
    t1 = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, row_order=ro, **kw))
    print(f"   conv {t0:.1f} us in table order, {t1:.1f} us in mask order; identical: {bool(torch.equal(ya, yb))}")
    for win in (256, 1024, 4096, 16384, 65536):  # mask order inside windows of consecutive rows only (keeps the gathers local)
        key = (torch.arange(nout, device=dev) // win << 27) | mask
        ow = torch.argsort(key, stable=True)
        row = ow.to(torch.int32)
        yw = ops.sparse_conv(x, w, tbl, nout, row_order=row, **({"wq": kw["wq"]} if kw else {}))
        if kw: ops.B3_VARIANT = 2
        tw = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, row_order=row, **kw))
        print(f"   windows of {win}: live per 16-row tile {tile_union(mask[ow], 16):.1f}, per 32 rows {tile_union(mask[ow], 32):.1f}; conv {tw:.1f} us; identical: {bool(torch.equal(ya, yw))}")
    ops.B3_VARIANT = 0
