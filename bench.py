"""Headline benchmark: input points/s of Pipeline.process_cloud on 1M-point synthetic trees.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; clouds are sharded, weak scaling)

A step = one pass of the hot path (CentreCloud -> blocks/voxelise -> UNet -> class filter ->
kNN graph -> components -> SSSP -> sample_tree -> prune/repair/smooth) over one 1M-point synthetic
tree, inputs resident in HBM when the timed region starts.  Clouds are independent, and a third of a
cloud's GPU time is spent in kernels that occupy ONE compute unit (the greedy branch selection) or wait for
the host to read a count back -- so every rank keeps `--streams` clouds in flight, each on its own host
thread and HIP stream; exactly K steps (clouds) are processed in the timed region.  Prints ONE JSON line
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


def auto_streams(steps: int, world: int) -> int:
    """Clouds in flight per rank.  Throughput saturates at 8 (the chip-filling kernels of 8 clouds hide each other's
    single-workgroup stages and host read-backs); every worker waits on its own stream with the runtime's
    spin-then-block policy, so leave two host cores per worker, and keep >= 3 clouds per worker so that the timed
    region does not end in a ragged tail."""
    by_cores = usable_cores() // (2 * max(world, 1))
    return max(1, min(8, steps // 3, by_cores))


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
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="clouds in flight per GPU (one host thread + HIP stream each); 0 = auto: 8, fewer when the run is "
                         "short (a worker should see >= 3 clouds) or the host has fewer than 2 cores per worker and rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # ST_BENCH_DRYRUN=1 (developer aid): exercise the multi-rank control flow on a box with ONE GPU -- every rank uses
    # cuda:0 and the result gather runs over gloo with host tensors.  Never set by the driver.
    dryrun = os.environ.get("ST_BENCH_DRYRUN") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if os.environ.get("ST_BENCH_SWITCH"):  # developer knob: the interpreter's GIL switch interval in seconds
        sys.setswitchinterval(float(os.environ["ST_BENCH_SWITCH"]))
    _sched = os.environ.get("ST_BENCH_SCHED")  # developer knob: how host threads wait for the GPU (spin | yield | block)
    if _sched:
        import ctypes
        _hip = ctypes.CDLL("libamdhip64.so")
        _rc = _hip.hipSetDeviceFlags({"spin": 1, "yield": 2, "block": 4}[_sched])
        print(f"hipSetDeviceFlags({_sched}) -> {_rc}", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = torch.device("cpu") if dryrun else device

    from smart_tree_amd import profiling
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.sharding import gather_skeletons, pack_skeleton
    from smart_tree_amd.synthetic import sample_tree_cloud

    import threading

    S = args.streams if args.streams > 0 else auto_streams(args.steps, world)
    pipes = [build_pipeline(device) for _ in range(S)]
    streams = [torch.cuda.Stream(device=device) for _ in range(S)]
    # two distinct clouds per rank, cycled; different seeds on every rank (independent trees)
    clouds = []
    for j in range(2):
        c = sample_tree_cloud(args.points, seed=rank * 2 + j)
        clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(device), rgb=torch.from_numpy(c["rgb"]).to(device)))

    finished = []  # packed skeletons of this rank, gathered to rank 0 once per timed region (no per-step rendezvous:
    #                 clouds differ in cost, a collective per step would make every step as slow as its slowest rank)
    last = {}

    def run_steps(total):
        """`total` steps, dealt to the S workers from a shared counter."""
        state = {"next": 0, "error": None}
        lock = threading.Lock()

        def worker(w):
            try:
                with torch.cuda.stream(streams[w]):
                    while True:
                        with lock:
                            i = state["next"]
                            state["next"] += 1
                        if i >= total:
                            break
                        sk = pipes[w].process_cloud(cloud=clouds[i % len(clouds)])
                        last["sk"] = sk
                        if world > 1:
                            finished.append(pack_skeleton(sk, cloud_id=rank * 1_000_000 + i))
                    streams[w].synchronize()
            except BaseException as e:  # noqa: BLE001 -- re-raised on the main thread
                state["error"] = e

        if S == 1:
            worker(0)
        else:
            threads = [threading.Thread(target=worker, args=(w,)) for w in range(S)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if state["error"] is not None:
            raise state["error"]

    def gather():
        if world > 1:
            gather_skeletons(finished, device=coll_device)
            finished.clear()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()  # inputs and weights are resident before any worker stream touches them
    warm = max(args.warmup, S) if args.warmup > 0 else 0  # every worker runs at least once before the clock starts
    run_steps(warm)
    gather()
    fence()
    # for the record (untimed): one cloud at a time on one stream = the latency of a single process_cloud call
    serial_ms = None
    if warm > 0:
        with torch.cuda.stream(streams[0]):
            t1 = time.perf_counter()
            for i in range(len(clouds)):
                pipes[0].process_cloud(cloud=clouds[i])
            streams[0].synchronize()
            serial_ms = 1e3 * (time.perf_counter() - t1) / len(clouds)
    fence()
    profiling.enable(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    gather()  # inside the timed region: the skeletons of all ranks end up on rank 0
    fence()
    dt = time.perf_counter() - t0
    profiling.enable(False)
    sk = last["sk"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
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
                       "clouds_per_rank": len(clouds), "parallelism": f"cloud-sharded x{world}",
                       "clouds_in_flight_per_gpu": S, "warmup_steps_run": warm,
                       "single_stream_ms_per_cloud": None if serial_ms is None else round(serial_ms, 3)},
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
