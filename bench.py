"""Headline benchmark: input points/s of Pipeline.process_cloud on 1M-point synthetic trees.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; clouds are sharded, weak scaling)

A step = one pass of the hot path (CentreCloud -> blocks/voxelise -> UNet -> class filter ->
kNN graph -> components -> SSSP -> sample_tree -> prune/repair/smooth) over one 1M-point synthetic
tree per rank, inputs resident in HBM when the timed region starts.  Prints ONE JSON line
(rank 0) with the throughput, the roofline of the dominant kernel measured live with HIP events,
and -- at N = 1 -- the CPU baseline (the oracle, i.e. a port of the reference algorithm: the
reference itself is CUDA-only) timed on this host's cores on one full cloud.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_POINTS = 1_000_000
VOXEL = 0.02
WEIGHTS = ROOT / "smart_tree_amd" / "model" / "weights" / "noble-elevator-58.npz"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_pipeline(device):
    from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
    from smart_tree_amd.model.model_inference import ModelInference
    from smart_tree_amd.pipeline import Pipeline
    from smart_tree_amd.skeleton.skeletonize import Skeletonizer

    mi = ModelInference("noble-elevator-58_model.pt", WEIGHTS, voxel_size=VOXEL, block_size=4, buffer_size=0.4, device=device)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=device)
    return Pipeline(AugmentationPipeline([CentreCloud()]), mi, sk, repair_skeletons=True, smooth_skeletons=True,
                    smooth_kernel_size=11, prune_skeletons=True, min_skeleton_radius=0.01, min_skeleton_length=0.02,
                    device=device)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(n_points: int):
    """The oracle pipeline on the host cores, one full cloud (bounded: ~30 s)."""
    from oracle import pipeline_oracle as po
    from oracle import unet_oracle as uo
    from smart_tree_amd.synthetic import sample_tree_cloud

    cores = usable_cores()
    torch.set_num_threads(cores)
    c = sample_tree_cloud(n_points, seed=0)
    w = uo.load_weights(WEIGHTS)
    timings = {}
    t0 = time.perf_counter()
    trees = po.process_cloud(c["xyz"], c["rgb"], w, VOXEL, timings=timings)
    dt = time.perf_counter() - t0
    return {"value": n_points / dt, "unit": "points/s", "cores": cores, "kind": "port",
            "sample": f"1 x {n_points}-point synthetic tree (seed 0), full pipeline, {dt:.1f} s; the graph stage of the "
                      "port is single-threaded C, the UNet uses torch-CPU on all cores",
            "stage_s": {k: round(v, 3) for k, v in timings.items()},
            "branches": int(sum(len(t.branches) for t in trees))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from smart_tree_amd import profiling
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.sharding import gather_skeletons, pack_skeleton
    from smart_tree_amd.synthetic import sample_tree_cloud

    pipe = build_pipeline(device)
    # two distinct clouds per rank, cycled; different seeds on every rank (independent trees)
    clouds = []
    for j in range(2):
        c = sample_tree_cloud(args.points, seed=rank * 2 + j)
        clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(device), rgb=torch.from_numpy(c["rgb"]).to(device)))

    def step(i, collect=None):
        sk = pipe.process_cloud(cloud=clouds[i % len(clouds)])
        if world > 1:
            gather_skeletons([pack_skeleton(sk, cloud_id=rank)], device=device)
        return sk

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    profiling.enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        sk = step(i)
    fence()
    dt = time.perf_counter() - t0
    profiling.enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * args.steps * args.points / dt
        out = {
            "metric": "points/sec end-to-end (voxelize->sparse-UNet->skeleton), 1M-pt tree",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: one {args.points}-point synthetic tree per rank per step, 2 cm voxels, "
                                   "noble-elevator-58 weights, full Pipeline.process_cloud with prune/repair/smooth",
                       "clouds_per_rank": len(clouds), "parallelism": f"cloud-sharded x{world}"},
            "roofline": profiling.roofline(HBM_PEAK_GBS),
            "stage_ms": profiling.stage_ms(args.steps),
            "last_result": {"trees": len(sk.skeletons), "branches": int(sum(len(t.branches) for t in sk.skeletons))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
