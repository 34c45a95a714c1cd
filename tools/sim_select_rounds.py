"""Developer aid: how many speculative rounds would the branch selection need under a given round scheme?  (CPU only.)
Builds the labelled cloud of a bench tree with the oracle network, graph + SSSP with the CPU sanitizer build of the kernels,
then replays sample_tree for the LARGEST component in tools/sim_select_rounds.c under several (entries, slots, mode) settings.
    python tools/sim_select_rounds.py [n_points] [seed]"""
import ctypes
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "hipemu"))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tree = Path(f"/tmp/sim_tree_{n}_{seed}.npz")
if not tree.exists():
    import build as emu_build  # noqa: E402

    from oracle import pipeline_oracle as po, unet_oracle as uo  # noqa: E402
    from smart_tree_amd import _lib  # noqa: E402
    from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
    from smart_tree_amd.skeleton import graph as G  # noqa: E402
    from smart_tree_amd.skeleton.filter import outlier_removal  # noqa: E402
    from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP, run_components  # noqa: E402
    from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

    cache = Path(f"/tmp/emu_lc_{n}_{seed}.npz")
    if cache.exists():
        z = np.load(cache)
        lc = {k: z[k] for k in z.files}
    else:
        c = sample_tree_cloud(n, seed=seed)
        w = uo.load_weights(ROOT / "smart_tree_amd" / "model" / "weights" / "noble-elevator-58.npz")
        lc = po.labelled_cloud(c["xyz"], c["rgb"], w, 0.02)
        np.savez(cache, **{k: v for k, v in lc.items() if isinstance(v, np.ndarray)})
    _lib._LIB = _lib.declare(ctypes.CDLL(str(emu_build.build())))
    _lib._ALLOW_HOST_POINTERS = True
    m = np.isin(lc["class_l"].reshape(-1), [0])
    bc = Cloud(xyz=torch.from_numpy(lc["xyz"][m].astype(np.float32)), medial_vector=torch.from_numpy(lc["medial_vector"][m].astype(np.float32)))
    medial, radius = G.medial_points(bc.xyz, bc.medial_vector)
    mask = outlier_removal(medial, radius.unsqueeze(1), 8)
    bc = bc.filter(mask)
    medial, radius = medial[mask], radius[mask]
    g = G.nn_graph(medial, radius.clamp(min=0.02), K=16)
    comps = g.connected_cugraph_components(32)
    res = run_components(comps, medial, radius, bc.xyz[:, 1].contiguous(), stages=STAGE_SSSP)
    off = comps.comp_off.numpy()
    sizes = np.diff(off)
    c = int(np.argmax(sizes))
    sl = slice(off[c], off[c + 1])
    order = comps.vert_order.long().numpy()[sl]
    np.savez(tree, pts=medial.numpy()[order].astype(np.float32), rad=radius.numpy()[order].astype(np.float32),
             pred=res.pred.numpy()[sl].astype(np.int64), dist=res.dist.numpy()[sl].astype(np.float32))
    print(f"{n} points seed {seed}: {comps.n_components} components, largest {sizes[c]} vertices -> {tree}", flush=True)

z = np.load(tree)
lib_path = Path("/tmp/libsimsel.so")
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", str(lib_path), str(ROOT / "tools" / "sim_select_rounds.c"), "-lm"], check=True)
L = ctypes.CDLL(str(lib_path))
P = ctypes.c_void_p
L.sim_rounds.argtypes = [ctypes.c_int64, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]
pts, rad, pred, dist = (np.ascontiguousarray(z[k]) for k in ("pts", "rad", "pred", "dist"))
nv = len(rad)
print(f"component of {nv} vertices")
print("  E   NS prune mode wpath | rounds slots commits big | ends: conflict wrong-guess exhausted bigcut | repairs branches cand(spec) cand(big) entries")
for E, NS, prune, mode, wpath, NSB in [(32, 16, 1.0, 0, 64, 0), (256, 16, 1.0, 1, 64, 0),
                                       (128, 16, 1.0, 4, 64, 16), (256, 16, 1.0, 4, 64, 16), (256, 16, 1.0, 4, 64, 32), (512, 16, 1.0, 4, 64, 32),
                                       (512, 16, 1.0, 4, 64, 1000), (1024, 16, 1.0, 4, 64, 1000), (256, 16, 1.5, 4, 64, 32), (256, 16, 0.7, 4, 64, 32),
                                       (512, 32, 1.0, 4, 64, 32), (1024, 32, 1.0, 4, 64, 64), (256, 16, 1.0, 4, 256, 16), (512, 16, 1.0, 4, 256, 32)]:
    stats = np.zeros(16, dtype=np.int64)
    L.sim_rounds(nv, pts.ctypes.data, rad.ctypes.data, pred.ctypes.data, dist.ctypes.data, E, NS, prune, mode, wpath, NSB,
                 stats.ctypes.data, None)
    print(f"{E:4d} {NS:4d}+{NSB:<3d} {prune:5.2f} {mode:4d} {wpath:5d} | {stats[0]:6d} {stats[1]:5d} {stats[2]:7d} {stats[3]:3d} | "
          f"{stats[4]:8d} {stats[5]:11d} {stats[6]:9d} {stats[11]:6d} | {stats[7]:7d} {stats[8]:8d} {stats[9]:10d} {stats[10]:9d} {stats[12]:7d} phaseB {stats[13]} late {stats[14]}", flush=True)
