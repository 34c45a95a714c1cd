"""CPU (tests/hipemu build of the kernel sources): rounds / slots / commits of the branch selection on a bench-like tree, for
trying selection parameters without a GPU.  The labelled cloud comes from the oracle network (cached in /tmp).
    python tools/emu_select_stats.py [n_points] [seed] ["k=v,..." skeleton params]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests" / "hipemu"))
import build as emu_build  # noqa: E402

from oracle import pipeline_oracle as po, unet_oracle as uo  # noqa: E402
from smart_tree_amd import _lib  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.skeleton import graph as G  # noqa: E402
from smart_tree_amd.skeleton.filter import outlier_removal  # noqa: E402
from smart_tree_amd.skeleton import tuning  # noqa: E402
from smart_tree_amd.skeleton.skeletonize import STAGE_SSSP, run_components  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
params = sys.argv[3] if len(sys.argv) > 3 else ""
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
L = _lib.lib()
m = np.isin(lc["class_l"].reshape(-1), [0])
bc = Cloud(xyz=torch.from_numpy(lc["xyz"][m].astype(np.float32)), medial_vector=torch.from_numpy(lc["medial_vector"][m].astype(np.float32)))
medial, radius = G.medial_points(bc.xyz, bc.medial_vector)
mask = outlier_removal(medial, radius.unsqueeze(1), 8)
bc = bc.filter(mask)
medial, radius = medial[mask], radius[mask]
g = G.nn_graph(medial, radius.clamp(min=0.02), K=16)
comps = g.connected_cugraph_components(32)
print(f"{n} points: graph vertices {len(bc)}, components {comps.n_components}, largest {comps.comp_size[:4].tolist()}", flush=True)
knobs = {}
for kv in filter(None, params.split(",")):
    k, v = kv.split("=")
    knobs[int(k)] = int(v)
ticks = torch.zeros(32, dtype=torch.int64)
t0 = time.time()
with tuning.override({**knobs, tuning.TICKS: ticks.data_ptr()}):
    res = run_components(comps, medial, radius, bc.xyz[:, 1].contiguous())
t = ticks.numpy()
print(f"params [{params}] ({time.time() - t0:.1f} s on the emulator): {res.stats}\n  rounds {t[8]} cached slots {t[13]} cached commits {t[12]} "
      f"evaluated by the replay {t[11]} | whole-workgroup iterations {t[9]} (path vertices {t[10]}) wide {t[14]} local {t[15]}\n"
      f"  rounds ended by: entries used up {t[16]}, a slot without a cache entry {t[17]}, an entry too long for the replay {t[18]}, too heavy {t[19]}")
