"""Dev script: per-layer timing of the sparse conv kernels (VALU vs MFMA) on a bench cloud's rulebooks.

    python tools/bench_conv.py [points=1000000] [voxel=0.02] [foliage_fraction=0]   # config 4: 5000000 0.01 0.6
"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
import torch, numpy as np
import bench
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.synthetic import sample_tree_cloud
from smart_tree_amd.dataset.dataset import voxelize_blocks
from smart_tree_amd.model import sparse_ops as ops
import ctypes as _ct, os as _os
from smart_tree_amd import _lib as _l
if _os.environ.get("ST_PROBE_LIB"):  # A/B aid: another build of the library (e.g. the previous revision's .so kept under _ab/)
    _l._LIB = _l.declare(_ct.CDLL(_os.environ["ST_PROBE_LIB"]))
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
NPTS = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
VOX = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
FOL = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # clouds per launch set (Cloud.collate): the batched pipeline's sizes
VARIANTS = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1]
B3_VARIANTS = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 and not sys.argv[6].startswith("--") else [1, 2]
clouds = []
for b in range(BATCH):
    c = sample_tree_cloud(NPTS, seed=(3 if FOL else 0) + b, **({"foliage_fraction": FOL} if FOL else {}))
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
cloud = pipe.preprocessing(Cloud.collate(clouds) if BATCH > 1 else clouds[0])
vb = voxelize_blocks(cloud.xyz, cloud.rgb, VOX, seg_off=cloud.seg_off)
coords0 = vb.coords
if "--input-order" not in sys.argv:  # as Smart_Tree.features does: Morton-ordered rows (st_spatial_order)
    coords0 = ops.move_rows(coords0, ops.spatial_order(coords0))
bricks = None if "--input-order" in sys.argv or "--hash" in sys.argv else ops.brick_pyramid(vb.coords, 3, vb.block_centres.shape[0], int(round(4.8 / VOX)) + 2, vb.blk_seg, vb.n_seg)
pyr = bricks[0] if bricks is not None else ops.build_pyramid(coords0, 3, vb.blk_seg, vb.n_seg)  # as Smart_Tree.features: brick order when it can
N = [x.shape[0] for x in pyr.coords]
print("levels", N)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
planes = [8, 16, 32, 64]
for lvl in range(4):
    C = planes[lvl]
    for (cin, cout, tbl, nout, name) in [(C, C, pyr.subm[lvl], N[lvl], "subm")] + \
            ([(2 * C, C, pyr.subm[lvl], N[lvl], "tail"), (C, 2 * C, pyr.down[lvl], N[lvl + 1], "down"), (2 * C, C, pyr.up[lvl], N[lvl], "up")] if lvl < 3 else []):
        nin = N[lvl + 1] if name == "up" else N[lvl]
        ro = pyr.up_order[lvl] if name == "up" else None  # the inverse convs run in parity order (as in model.py)
        x = torch.randn(nin, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        pairs = int((tbl >= 0).sum())
        bytes_ = pairs * (cin * 4 + 4) + nout * cout * 4
        flops = 2 * pairs * cin * cout
        t_valu = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, row_order=ro))
        line = f"L{lvl} {name:5s} {cin:3d}->{cout:3d} N={nout:7d} P={pairs:8d}: VALU {t_valu:7.1f} us ({bytes_/t_valu/1e3:7.1f} GB/s = {bytes_/t_valu/1e3/80:4.1f} %, {flops/t_valu/1e6:6.2f} TF)"
        if cin % 16 == 0 and cout % 16 == 0:
            wp = ops.mfma_weight(w)
            ya = ops.sparse_conv(x, w, tbl, nout, row_order=ro)
            for var, tag in [(v, f"v{v}") for v in VARIANTS]:
                ops.MFMA_VARIANT = var  # passed per call (st_sparse_conv_mfma_fwd's `variant`)
                t_m = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, wp=wp, row_order=ro))
                yb = ops.sparse_conv(x, w, tbl, nout, wp=wp, row_order=ro)
                err = (ya - yb).abs().max().item() / (ya.abs().max().item() + 1e-30)
                line += f" | mfma[{tag}] {t_m:6.1f} us ({bytes_/t_m/1e3:7.1f} GB/s = {bytes_/t_m/1e3/80:4.1f} % of 8 TB/s) {flops/t_m/1e6:5.1f} TF e={err:.0e}"
            ops.MFMA_VARIANT = 0
            if ops.b3_eligible(cin, cout, cin):  # split-bf16 matrix-core kernel (float32 features, bf16 pipe), 1 / 2 row tiles per wave
                wq = ops.b3_weight(w)
                y64 = None
                for var in B3_VARIANTS:
                    ops.B3_VARIANT = var
                    t_b = timeit(lambda: ops.sparse_conv(x, w, tbl, nout, wq=wq, row_order=ro))
                    yq = ops.sparse_conv(x, w, tbl, nout, wq=wq, row_order=ro)
                    errq = (ya - yq).abs().max().item() / (ya.abs().max().item() + 1e-30)
                    line += f" | b3[rt{var}] {t_b:6.1f} us ({bytes_/t_b/1e3:7.1f} GB/s = {bytes_/t_b/1e3/80:4.1f} %) {flops/t_b/1e6:5.1f} TF e={errq:.0e}"
                ops.B3_VARIANT = 0
            # half-precision storage (config 5): f16 matrix-core kernel, half the gather bytes
            xh, wph = x.half(), ops.mfma_weight16_half(w)
            bytes_h = pairs * (cin * 2 + 4) + nout * cout * 2
            t_h = timeit(lambda: ops.sparse_conv(xh, w, tbl, nout, out_half=True, wp16=wph, row_order=ro))
            yh = ops.sparse_conv(xh, w, tbl, nout, out_half=True, wp16=wph, row_order=ro).float()
            errh = (ya - yh).abs().max().item() / (ya.abs().max().item() + 1e-30)
            line += f" | f16 {t_h:6.1f} us ({bytes_h/t_h/1e3:7.1f} GB/s = {bytes_h/t_h/1e3/80:4.1f} %) {flops/t_h/1e6:5.1f} TF e={errh:.0e}"
        print(line)
