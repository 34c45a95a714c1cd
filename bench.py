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


class CloudWorker:
    """S pipelines on S HIP streams of one process, two resident 1M-point clouds, steps dealt from a shared counter."""

    def __init__(self, device, n_streams, n_points, rank):
        from smart_tree_amd.data_types.cloud import Cloud
        from smart_tree_amd.synthetic import sample_tree_cloud

        self.device, self.S, self.rank = device, n_streams, rank
        self.pipes = [build_pipeline(device) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        # two distinct clouds per rank, cycled; different seeds on every rank (independent trees)
        self.clouds = []
        for j in range(2):
            c = sample_tree_cloud(n_points, seed=rank * 2 + j)
            self.clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(device), rgb=torch.from_numpy(c["rgb"]).to(device)))
        self.last = None
        self.id_base = 0

    def run(self, total, collect):
        """`total` steps, dealt to the S worker threads from a shared counter; returns the packed skeletons (if asked
        for) once every stream has drained."""
        import threading

        from smart_tree_amd.sharding import pack_skeleton

        state = {"next": 0, "error": None}
        lock = threading.Lock()
        finished = []

        def work(w):
            try:
                with torch.cuda.stream(self.streams[w]):
                    while True:
                        with lock:
                            i = state["next"]
                            state["next"] += 1
                        if i >= total:
                            break
                        sk = self.pipes[w].process_cloud(cloud=self.clouds[i % len(self.clouds)])
                        self.last = sk
                        if collect:
                            finished.append(pack_skeleton(sk, cloud_id=self.rank * 1_000_000 + self.id_base + i))
                    self.streams[w].synchronize()
            except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                state["error"] = e

        if self.S == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(w,)) for w in range(self.S)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if state["error"] is not None:
            raise state["error"]
        return finished

    def serial_ms(self):
        """Untimed, for the record: one cloud at a time on one stream = the latency of a single process_cloud call."""
        with torch.cuda.stream(self.streams[0]):
            t1 = time.perf_counter()
            for cloud in self.clouds:
                self.pipes[0].process_cloud(cloud=cloud)
            self.streams[0].synchronize()
            return 1e3 * (time.perf_counter() - t1) / len(self.clouds)

    def report(self, steps):
        from smart_tree_amd import profiling

        sk = self.last
        return {"roofline": profiling.roofline(HBM_PEAK_GBS), "stage_ms": profiling.stage_ms(steps),
                "last_result": {"trees": len(sk.skeletons), "branches": int(sum(len(t.branches) for t in sk.skeletons))}}


def _proc_worker(conn, local_rank, n_streams, n_points, rank, widx):
    """--procs helper: owns its own HIP context, pipelines and clouds; obeys run / serial / report / exit."""
    from smart_tree_amd import profiling

    torch.cuda.set_device(local_rank)
    w = CloudWorker(torch.device("cuda", local_rank), n_streams, n_points, rank)
    w.id_base = (widx + 1) * 10_000
    torch.cuda.synchronize()
    conn.send(("ready", None))
    while True:
        msg = conn.recv()
        if msg[0] == "run":
            profiling.enable(bool(msg[3]))
            out = w.run(msg[1], msg[2])
            torch.cuda.synchronize()
            conn.send(("done", [(t.numpy(), g.numpy()) for t, g in out]))
        elif msg[0] == "serial":
            conn.send(("serial", w.serial_ms()))
        elif msg[0] == "report":
            conn.send(("report", w.report(msg[1])))
        else:
            break


def _expect(conn, tag, timeout_s):
    if not conn.poll(timeout_s):
        raise RuntimeError(f"bench helper process did not answer '{tag}' within {timeout_s} s")
    got = conn.recv()
    assert got[0] == tag, got
    return got[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--procs", type=int, default=1,
                    help="helper processes per GPU, --streams clouds in flight each (opt-in; the default keeps the one "
                         "process per GPU of the launch contract)")
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
    from smart_tree_amd.sharding import gather_skeletons

    P = max(1, args.procs)
    S = args.streams if args.streams > 0 else auto_streams(args.steps // P, world * P)
    finished = []  # packed skeletons of this rank, gathered to rank 0 once per timed region (no per-step rendezvous:
    #                 clouds differ in cost, a collective per step would make every step as slow as its slowest rank)
    if P == 1:
        worker = CloudWorker(device, S, args.points, rank)
        children = []

        def run_steps(total):
            finished.extend(worker.run(total, collect=world > 1))
    else:
        # --procs P (opt-in): P helper processes drive this rank's GPU, S streams each; this process
        # keeps the rank's place in the process group, hands out the steps and does the result gather
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        children = []
        for w in range(P):
            here, there = ctx.Pipe()
            proc = ctx.Process(target=_proc_worker, args=(there, local_rank, S, args.points, rank, w), daemon=True)
            proc.start()
            children.append((proc, here))
        for _, conn in children:
            _expect(conn, "ready", 600)
        worker = None

        def run_steps(total):
            for w, (_, conn) in enumerate(children):
                conn.send(("run", total // P + (1 if w < total % P else 0), world > 1, profiling.enabled()))
            for _, conn in children:
                finished.extend((torch.from_numpy(t), torch.from_numpy(g)) for t, g in _expect(conn, "done", 600))

    def gather():
        if world > 1:
            gather_skeletons(finished, device=coll_device)
            finished.clear()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()  # inputs and weights are resident before any worker stream touches them
    warm = max(args.warmup, S * P) if args.warmup > 0 else 0  # every worker runs at least once before the clock starts
    run_steps(warm)
    gather()
    fence()
    # for the record (untimed): one cloud at a time on one stream = the latency of a single process_cloud call
    serial_ms = None
    if warm > 0:
        if worker is not None:
            serial_ms = worker.serial_ms()
        else:
            children[0][1].send(("serial",))
            serial_ms = _expect(children[0][1], "serial", 600)
    fence()
    profiling.enable(True)
    t0 = time.perf_counter()
    run_steps(args.steps)  # returns when every worker has synchronised its streams
    gather()  # inside the timed region: the skeletons of all ranks end up on rank 0
    fence()
    dt = time.perf_counter() - t0
    profiling.enable(False)
    if worker is not None:
        report = worker.report(args.steps)
    else:  # the roofline / stage times of helper 0 (each helper measures its own launches with HIP events)
        children[0][1].send(("report", args.steps // P + (1 if args.steps % P else 0)))
        report = _expect(children[0][1], "report", 600)
        for proc, conn in children:
            conn.send(("exit",))
        for proc, _ in children:
            proc.join(30)
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
                       "clouds_per_rank": 2, "parallelism": f"cloud-sharded x{world}",
                       "clouds_in_flight_per_gpu": S * P, "worker_processes_per_gpu": P, "warmup_steps_run": warm,
                       "single_stream_ms_per_cloud": None if serial_ms is None else round(serial_ms, 3)},
            "roofline": report["roofline"],
            "stage_ms": report["stage_ms"],
            "last_result": report["last_result"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.points)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
