"""GPU box: configs[3] (5M-point dense canopy, 1 cm) as the batch of two clouds bench.py runs, under a list of skeleton tuning
settings: wall ms per cloud, stage brackets, the skeleton kernels' totals.   python tools/probe_canopy.py ["k=v,.." ...]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from smart_tree_amd import profiling  # noqa: E402
from smart_tree_amd.data_types.cloud import Cloud  # noqa: E402
from smart_tree_amd.skeleton import tuning  # noqa: E402
from smart_tree_amd.synthetic import sample_tree_cloud  # noqa: E402

dev = torch.device("cuda:0")
nb = 2
clouds = []
for b in range(nb):
    c = sample_tree_cloud(5_000_000, seed=3 + b, foliage_fraction=0.6)
    clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev)))
pipe = bench.build_pipeline(dev, voxel=0.01)
pipe.process_clouds(clouds)
for params in (sys.argv[1:] or [""]):
    knobs = {}
    for kv in filter(None, params.split(",")):
        k, v = kv.split("=")
        knobs[int(k)] = int(v)
    with tuning.override(knobs):
        for sub in (clouds, clouds[:1]):
            pipe.process_clouds(sub)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                pipe.process_clouds(sub)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 2
            profiling.enable(True)
            pipe.process_clouds(sub)
            torch.cuda.synchronize()
            profiling.enable(False)
            roof = profiling.roofline(8000.0) or {}
            st = profiling.stage_ms(1)
            ak = roof.get("all_kernels") or {}
            sk = {k.split("(")[0]: (v["total_ms"], v["launches"]) for k, v in ak.items() if k.startswith("k_sk_")}
            print(f"params [{params}] {len(sub)} cloud(s): {ms:.1f} ms per launch set; skeleton_kernels {st.get('skeleton_kernels')} ms; {sk}", flush=True)
