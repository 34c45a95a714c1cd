"""Headline benchmark: input points/s of the smart-tree inference path on 1M-point synthetic trees.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; clouds are sharded, weak scaling -- every rank runs K clouds of its own;
     --scaling strong --clouds 64: BASELINE.json configs[2] as written, a FIXED batch split round-robin over the ranks)

A step = one pass of the hot path (CentreCloud -> blocks/voxelise -> UNet -> class filter -> kNN graph -> components
-> SSSP -> sample_tree -> prune/repair/smooth -> host copy of the skeleton) over ONE 1M-point synthetic tree
(BASELINE.json configs[1]), inputs resident in HBM when the timed region starts (the bench contract: the PCIe-inclusive rate is
reported beside it, never as `value`).  Clouds are independent, so a rank
runs them in BATCHES of up to `--batch` clouds through one launch set (`Pipeline.process_clouds`: the batch index is
carried through every kernel; results per cloud are bit-identical to one cloud at a time, tests/test_batch.py) and keeps
`--streams` batches in flight (one host thread + HIP stream each), so that the single-workgroup skeleton stages of one
batch overlap the chip-wide kernels of the other.  Every step of a pass is a DIFFERENT cloud (seeds rank * D .. rank * D
+ D - 1, D = min(K, 64): BASELINE configs[2] names seeds 0..63).

Timed region: PASSES (5) passes of exactly K steps, each bracketed by barrier + synchronize on both sides; `value` is the
MEDIAN pass (min / max / all passes are in `passes`).  Prints ONE JSON line (rank 0) with, beside the contract's fields:
  roofline                  the kernel class with the largest summed launch duration in the timed region (HIP events on the
                            launch stream), priced against the HBM peak
  roofline_gather_scatter   the sparse-convolution family (gather / rule-GEMM / scatter) of the same timed region
  single_cloud              SURVEY 8d's literal metric: median wall time of >= 5 `Pipeline.process_cloud(cloud=...)` calls
                            with the cloud in PINNED HOST memory (the upload is inside the call), per-stage ms, and the
                            convolution family's roofline for one cloud at a time
  value_incl_host_upload    the same K steps with every cloud uploaded from pinned host memory inside the pass (copy stream,
                            the next batch's upload overlaps the current batch's kernels); median of 3 passes
  representative_e2e        (N = 1) the same K clouds through the whole path with the skeleton stage fed the GENERATOR's exact medial
                            vectors (the network still runs and is timed): one end-to-end figure on the regime the reference runs
                            the skeleton stage in -- the shipped checkpoint's output on the benchmark tree is arbitrary and gives a
                            2-3x cheaper graph; the first cloud's skeleton is compared with the oracle's in the run
  config.strong_scaling_projection  (N = 1) launch sets of 8 / 16 / 32 / 64 of configs[2]'s 64 clouds timed on this one GPU and the
                            speed-up eight ranks with 8 clouds each would reach: a PROJECTION, not a scaling measurement
  cpu_baseline              (N = 1) the oracle -- a port of the reference algorithm; the reference itself is CUDA-only --
                            timed on this host's cores on one full cloud, whose skeleton is also compared with the GPU's
                            for the same cloud (`parity_in_run`).

Environment (all optional): ST_BENCH_ORDERED 3 (default: batches in flight take turns with voxelise .. network) / 1 (with the
whole chip-filling phase) / 0 (free-running); ST_BENCH_MIN_UPTIME_S (30: the warm-up lasts until the process is that old);
ST_BENCH_BLOCKING_SYNC (1: host waits block instead of spinning; default: by the number of host cores); ST_BENCH_TORCH_THREADS (1); ST_BENCH_SELECT_THREADS,
ST_SKELETON_PARAMS (developer knobs); ST_BENCH_DRYRUN=1 (multi-rank control flow on one GPU over gloo).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

# A bench that starts right after another GPU process has exited -- the round-end sequence pytest -> smoke -> bench -- runs in a
# slow state for its first ~15-20 s on the gpurun boxes (profiles/r02_after_another_process.txt): 2.5-2.9 ms per cloud instead of
# 1.30, every stage that ends in a count read-back 5-10x longer; an idle probe of such a read-back shows nothing (20 us).
# HSA_ENABLE_SDMA=0 (host copies by shader kernels) shortens it, waiting it out removes it: the warm-up lasts until the process
# is MIN_UPTIME seconds old (with either copy path 1.29-1.30 ms per cloud then; the SDMA engines stay on: faster uploads).
T_PROCESS = time.perf_counter()
MIN_UPTIME = float(os.environ.get("ST_BENCH_MIN_UPTIME_S", "30"))  # warm-up lasts at least until the process is this old

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_POINTS = 1_000_000
VOXEL = 0.02
WEIGHTS = ROOT / "smart_tree_amd" / "model" / "weights"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MAX_BATCH = 64  # clouds per launch set  } measured on one MI355X (tools/sweep_batch.sh, profiles/r02_sweep_batch.txt, 384 steps, final kernels):
STREAMS = 2     # batches in flight      } phases in turns: voxelise .. network 2 x 64 = 1.29 ms per cloud (3 x 64 the same); voxelise .. adjacency 2 x 64 = 1.32;
FREE_STREAMS = 3  #                       } free-running: 3 x 64 = 1.25-1.27, 3 x 48 = 1.30, 2 x 64 = 1.29, 4 x 32 = 1.29, 4 x 48 = 1.32
MAX_DISTINCT = int(os.environ.get("ST_BENCH_DISTINCT", "64"))  # distinct clouds per rank (BASELINE configs[2]: seeds 0..63); fewer steps -> one
#                                                              cloud per step.  ST_BENCH_DISTINCT=4: round 3's four cycled clouds (comparison aid)
PASSES = 5         # timed passes of K steps; the median is reported
UPLOAD_PASSES = 3
SINGLE_CALLS = 7   # process_cloud calls of the single-cloud block
REP_SET = 20       # clouds of the ground-truth-medial launch set (representative_inputs)
PROJECTION_CLOUDS = 64  # BASELINE.json configs[2]: a fixed batch of 64 trees (seeds 0..63) over 8 GPUs
ORDERED = int(os.environ.get("ST_BENCH_ORDERED", "3"))  # 3 (default): voxelise .. network of the batches in flight take turns; 1: voxelise .. adjacency; 0: free-running; 2: only the conv sequences
SHORT_RUN_STEPS = 128  # below this a run is one or two rounds of batches: nothing to take turns with, the batches run free (unless ST_BENCH_ORDERED is set)


def build_pipeline(device, weights="noble-elevator-58", voxel=VOXEL, fp16=False, blocking="blocks"):
    from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
    from smart_tree_amd.model.model_inference import ModelInference
    from smart_tree_amd.pipeline import Pipeline
    from smart_tree_amd.skeleton.skeletonize import Skeletonizer

    mi = ModelInference(f"{weights}_model.pt", WEIGHTS / f"{weights}.npz", voxel_size=voxel, block_size=4, buffer_size=0.4,
                        device=device, fp16=fp16, blocking=blocking)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=device)
    sk.block_threads = int(os.environ.get("ST_BENCH_SELECT_THREADS", "0"))  # developer knob: lanes of the per-tree selection workgroup
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


def plan_batches(steps: int, streams: int, max_batch: int):
    """K clouds -> batch sizes: as few batches as possible, a multiple of the stream count (no ragged tail: every
    stream gets the same number of equally sized batches where K allows it)."""
    if steps <= 0:
        return []
    forced = os.environ.get("ST_BENCH_PLAN")  # developer knob (tools/sweep_r3_plan.sh): explicit batch sizes, e.g. "8,6,4,2"
    if forced:
        sizes = [int(v) for v in forced.split(",")]
        if sum(sizes) == steps:
            return sizes
    # small batches make poor use of a fourth host thread (the Python side of a batch is ~constant): measured on one MI355X,
    # 20 clouds: 1 / 2 / 3 / 4 in flight = 2.72 / 2.61 / 2.60 / 3.67 ms per cloud; 48 clouds: 2 / 3 / 4 = 1.99 / 1.96 / 2.39
    streams = max(1, min(streams, steps // 16), min(streams, 3, steps // 6))
    per_round = streams * max_batch
    n_batches = streams * ((steps + per_round - 1) // per_round)
    n_batches = min(n_batches, steps)
    if steps <= 32 and steps <= max_batch:
        # a short run is ONE batch: with the clouds of a pass in one launch set the chip-filling kernels run at their best and
        # the skeleton stage (one compute unit per tree + its helper workgroups, ~12 ms whatever the batch size) is exposed once.
        # Measured at the driver's --steps 20 (round 4): 20 = 1.60 ms per step, 12 + 8 on two streams 1.62-1.66, 10 + 10 1.57-1.73.
        return [steps]
    base, extra = divmod(steps, n_batches)
    return [base + (1 if i < extra else 0) for i in range(n_batches)]


def literal_query_seconds(xyz, medial_vector, class_l, workers=-1):
    """BASELINE.md section 2's LITERAL branch query, timed: for every branch the reference's `select_path_points`
    (skeleton/path.py:19-46) asks for the nearest path vertex of ALL points of the component (`nn(points, path_points, r =
    max path radius)`, K = 1) -- here scipy's cKDTree over the path, queried with every component point, all cores.  The port
    (oracle/skeleton_oracle.c so_sample_tree) inverts the query (path vertices look up the points around them in a grid), like
    the GPU path; both give the same claimed sets.  Returns (seconds of the literal queries, seconds of the port's sample_tree
    on the same components, branches)."""
    from scipy.spatial import cKDTree

    from oracle import skeleton_oracle as so

    sel = np.isin(class_l.reshape(-1), (0,))
    sk = so.skeletonize(xyz[sel], medial_vector[sel])
    pts_all = (xyz[sel] + medial_vector[sel])[np.nonzero(sk.keep_mask)[0]].astype(np.float32)
    mv = medial_vector[sel][np.nonzero(sk.keep_mask)[0]]
    rad_all = np.sqrt((mv * mv).sum(1)).astype(np.float32)
    t_lit = t_port = 0.0
    n_br = 0
    for comp in sk.components:
        pts, rad = pts_all[comp.vertex_ids], rad_all[comp.vertex_ids]
        t0 = time.perf_counter()
        so.sample_tree(pts, rad, comp.preds, comp.tree_dist)
        t_port += time.perf_counter() - t0
        t0 = time.perf_counter()
        for b in comp.branches:
            path = pts[b.verts]
            r = float(rad[b.verts].max())
            d, i = cKDTree(path).query(pts, k=1, distance_upper_bound=r, workers=workers)
            hit = np.isfinite(d)
            _ = hit & (d < np.where(hit, rad[b.verts][np.minimum(i, len(path) - 1)], 0.0))  # path.py:35-40
            n_br += 1
        t_lit += time.perf_counter() - t0
    return t_lit, t_port, n_br


def cpu_baseline(n_points: int):
    """The oracle pipeline on the host cores, one full cloud (bounded: ~10 s).  Returns (json entry, labelled cloud)."""
    from oracle import pipeline_oracle as po
    from oracle import unet_oracle as uo
    from smart_tree_amd.synthetic import sample_tree_cloud

    cores = usable_cores()
    torch.set_num_threads(cores)
    c = sample_tree_cloud(n_points, seed=0)
    w = uo.load_weights(WEIGHTS / "noble-elevator-58.npz")
    timings = {}
    t0 = time.perf_counter()
    lc = po.labelled_cloud(c["xyz"], c["rgb"], w, VOXEL, timings=timings)
    t1 = time.perf_counter()
    trees = po.skeleton_from_labelled(lc["xyz"], lc["medial_vector"], lc["class_l"])
    t2 = time.perf_counter()
    po.post_process(trees)
    dt = time.perf_counter() - t0
    timings["skeleton"], timings["post_process"] = t2 - t1, time.perf_counter() - t2
    t_lit, t_port, n_br = literal_query_seconds(lc["xyz"], lc["medial_vector"], lc["class_l"])
    dt_literal = dt - t_port + t_lit
    entry = {"value": n_points / dt, "unit": "points/s", "cores": cores, "kind": "port",
             "note": "port of the reference algorithm with an INVERTED branch query (path vertices look up the points around them) and a "
                     "single-threaded C graph stage; it is NOT the reference's CPU path (the reference has none: spconv / FRNN / cugraph "
                     "are CUDA-only).  `value_literal_query` is the same run with the branch selection's time replaced by "
                     "BASELINE.md section 2's literal all-points nearest-path-vertex query per branch (scipy cKDTree, workers = all cores)",
             "value_literal_query": n_points / dt_literal,
             "literal_query": {"seconds_literal_queries": round(t_lit, 3), "seconds_port_sample_tree": round(t_port, 3),
                               "branches": n_br, "seconds_whole_path": round(dt_literal, 2)},
             "sample": f"1 x {n_points}-point synthetic tree (seed 0), full pipeline, {dt:.1f} s; the graph stage of the "
                       "port is single-threaded C, the UNet uses torch-CPU on all cores",
             "stage_s": {k: round(v, 3) for k, v in timings.items()},
             "branches": int(sum(len(t.branches) for t in trees))}
    return entry, lc


def parity_in_run(pipe, cloud, cpu_lc):
    """GPU result for the SAME cloud as the CPU leg (seed 0), outside the timed region:
    (a) labelled cloud: voxel representatives identical, medial vectors within the fp32 bar, classes equal on > 99.9 %;
    (b) skeleton: the oracle's skeleton + post-processing of the GPU's labelled cloud equals the GPU's, branch by branch
        (ids, parents, vertex coordinates, radii -- exact)."""
    from oracle import pipeline_oracle as po

    sk = pipe.process_cloud(cloud=cloud)
    lc = pipe.last_labelled_cloud
    xyz, mv, cls = lc.xyz.cpu().numpy(), lc.medial_vector.cpu().numpy(), lc.class_l.cpu().numpy()
    ok_xyz = bool(np.array_equal(xyz, cpu_lc["xyz"]))
    scale = float(np.sqrt(np.mean(cpu_lc["medial_vector"].astype(np.float64) ** 2))) + 1e-30
    mv_err = float(np.abs(mv - cpu_lc["medial_vector"]).max() / scale) if ok_xyz else float("nan")
    cls_diff = float((cls != cpu_lc["class_l"]).mean()) if ok_xyz else float("nan")
    trees = po.skeleton_from_labelled(xyz, mv, cls)
    po.post_process(trees)
    same = len(trees) == len(sk.skeletons)
    n_br = 0
    if same:
        for got, ref in zip(sk.skeletons, trees):
            same = same and list(got.branches) == list(ref.branches)
            if not same:
                break
            for k, rb in ref.branches.items():
                gb = got.branches[k]
                same = same and gb.parent_id == rb.parent_id and np.array_equal(gb.xyz.numpy(), rb.xyz) and \
                    np.array_equal(gb.radii.numpy(), rb.radii)
            n_br += len(ref.branches)
    # the CPU leg's network is float32 torch-CPU: with this checkpoint's BatchNorm statistics two float32 evaluation orders
    # differ by ~1e-2 of the output's rms (tests/ hold the HIP path to the float64 oracle: 1e-4 or 4x the float32 oracle's own
    # distance); here the bar is 3e-2 against that float32 leg
    return {"parity_in_run": bool(same and ok_xyz and mv_err < 3e-2 and cls_diff < 1e-3), "skeleton_identical": bool(same),
            "branches_compared": n_br, "labelled_points_identical": ok_xyz, "medial_vector_rel_err_vs_cpu_fp32": mv_err,
            "class_mismatch_fraction": cls_diff}


def _gen_cloud(args):
    """Pool worker: one synthetic cloud into the shared host buffer (no GPU is touched in the children)."""
    n_points, seed, j = args
    from smart_tree_amd.synthetic import sample_tree_cloud

    c = sample_tree_cloud(n_points, seed=seed)
    _SHARED[j].numpy()[:] = c["xyz"]
    if _SHARED_MV is not None and j < _SHARED_MV.shape[0]:
        _SHARED_MV[j].numpy()[:] = c["medial_vector"]
    return j


_SHARED = None
_SHARED_MV = None


def generate_clouds(n_points: int, seeds, procs: int, n_mv: int = 0):
    """The synthetic clouds of this rank: [D, n, 3] float32 host tensor, generated by `procs` forked processes BEFORE the
    process has a GPU context (1.4 s of numpy per 1M-point cloud on one core; rgb is all zeros in the generator).  n_mv: the
    generator's exact medial vectors of the first n_mv clouds are kept as well ([n_mv, n, 3]; `representative_inputs`)."""
    global _SHARED, _SHARED_MV
    import multiprocessing as mp

    _SHARED = torch.empty((len(seeds), n_points, 3), dtype=torch.float32).share_memory_()
    _SHARED_MV = torch.empty((min(n_mv, len(seeds)), n_points, 3), dtype=torch.float32).share_memory_() if n_mv > 0 else None
    jobs = [(n_points, int(s), j) for j, s in enumerate(seeds)]
    if procs <= 1 or len(jobs) == 1:
        for job in jobs:
            _gen_cloud(job)
    else:
        with mp.get_context("fork").Pool(min(procs, len(jobs))) as pool:
            list(pool.imap_unordered(_gen_cloud, jobs))
    out, mv, _SHARED, _SHARED_MV = _SHARED, _SHARED_MV, None, None
    return (out, mv) if n_mv > 0 else out


class CloudWorker:
    """S pipelines on S HIP streams of one process; resident 1M-point clouds; batches dealt from a shared list."""

    def __init__(self, device, n_streams, host_xyz, rank):
        from smart_tree_amd.data_types.cloud import Cloud

        self.device, self.S, self.rank = device, n_streams, rank
        self.pipes = [build_pipeline(device) for _ in range(n_streams)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        self.copy_streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
        n_points = host_xyz.shape[1]
        zeros = torch.zeros((n_points, 3), dtype=torch.float32).pin_memory()  # the generator's rgb: one host buffer serves every cloud
        self.host, self.clouds = [], []
        for j in range(host_xyz.shape[0]):
            xyz = host_xyz[j].pin_memory()
            self.host.append((xyz, zeros))
            self.clouds.append(Cloud(xyz=xyz.to(device), rgb=zeros.to(device)))
        self.last = None
        self.first = 0  # offset into the resident clouds (strong_scaling_projection times other eighths of the 64)

    def batch_clouds(self, first, size):
        return [self.clouds[(self.first + first + k) % len(self.clouds)] for k in range(size)]

    def upload_batch(self, w, first, size, after=None):
        """Host -> device copies of one batch on worker w's COPY stream (pinned buffers); returns (clouds, event).  `after`: the
        event of the previous batch's copies -- the batches of a pass go over PCIe one after the other, in batch order, each at the
        full rate (two batches copying side by side both arrive late)."""
        from smart_tree_amd.data_types.cloud import Cloud

        cs = self.copy_streams[w]
        if after is not None:
            cs.wait_event(after)
        with torch.cuda.stream(cs):
            out = []
            for k in range(size):
                xyz, rgb = self.host[(first + k) % len(self.host)]
                out.append(Cloud(xyz=xyz.to(self.device, non_blocking=True), rgb=rgb.to(self.device, non_blocking=True)))
            ev = torch.cuda.Event()
            ev.record(cs)
        return out, ev

    def run(self, batches, collect, upload=False, streams=None, mode=None):
        """`batches`: list of batch sizes, dealt to the worker threads from a shared counter.  streams: worker threads / HIP
        streams to use (default: all); mode: ORDERED for this call (default: the module setting)."""
        S = self.S if streams is None else max(1, min(int(streams), self.S))
        mode = ORDERED if mode is None else mode
        from smart_tree_amd.sharding import pack_skeleton

        starts = np.concatenate([[0], np.cumsum(batches)]).tolist()
        state = {"next": 0, "error": None, "wide_done": None}
        lock = threading.Lock()
        up_cond, up_state = threading.Condition(), {"next": 0, "event": None}
        finished = []
        # Ordered chip-filling phases (ST_BENCH_ORDERED=1): the batches in flight take turns with the part of the pipeline whose
        # kernels fill the chip (voxelise .. adjacency): two such kernels gain nothing from sharing it, they only make each
        # other's launches last longer.  What overlaps with a batch's chip-filling phase is the others' skeleton stage (one
        # compute unit per tree).  `wide` is held from the start of a batch to Skeletonizer.on_wide_phase_done; the event makes
        # the order hold on the GPU as well.
        ordered = mode in (1, 3) and S > 1
        conv_ordered = mode == 2 and S > 1
        wide = threading.Lock()

        import contextlib

        def conv_gate_for(w):
            @contextlib.contextmanager
            def gate():
                with wide:
                    if state["wide_done"] is not None:
                        self.streams[w].wait_event(state["wide_done"])
                    try:
                        yield
                    finally:
                        ev = torch.cuda.Event()
                        ev.record(self.streams[w])
                        state["wide_done"] = ev
            return gate

        for w in range(self.S):
            self.pipes[w].model_inference.model.conv_gate = conv_gate_for(w) if conv_ordered else None

        def work(w):
            try:
                torch.cuda.set_device(self.device)  # the current device is thread-local: a new thread starts on device 0
                with torch.cuda.stream(self.streams[w]):
                    # upload mode: worker w owns batches w, w + S, ..: the NEXT one's host -> device copies are enqueued on the
                    # copy stream before this one's kernels, so they overlap (the first upload of a worker overlaps the other
                    # workers' kernels only)
                    mine = list(range(w, len(batches), S)) if upload else None

                    def upload_in_turn(i):  # batch i's copies are enqueued after batch i - 1's (host side) and run after them (event)
                        with up_cond:
                            while up_state["next"] != i and state["error"] is None:
                                up_cond.wait(timeout=0.05)
                            got = self.upload_batch(w, starts[i], batches[i], after=up_state["event"])
                            up_state["next"], up_state["event"] = i + 1, got[1]
                            up_cond.notify_all()
                        return got

                    pending = upload_in_turn(mine[0]) if upload and mine else None
                    turn = 0
                    while True:
                        if upload:
                            if turn >= len(mine):
                                break
                            i = mine[turn]
                            turn += 1
                            clouds, ev = pending
                            pending = upload_in_turn(mine[turn]) if turn < len(mine) else None
                            self.streams[w].wait_event(ev)
                            for cl in clouds:  # allocated on the copy stream, used on this one
                                cl.xyz.record_stream(self.streams[w])
                                cl.rgb.record_stream(self.streams[w])
                        else:
                            with lock:
                                i = state["next"]
                                state["next"] += 1
                            if i >= len(batches):
                                break
                            clouds = self.batch_clouds(starts[i], batches[i])
                        if ordered:
                            wide.acquire()
                            held = [True]
                            if state["wide_done"] is not None:
                                self.streams[w].wait_event(state["wide_done"])

                            def release(held=held, w=w):
                                if held[0]:
                                    ev = torch.cuda.Event()
                                    ev.record(self.streams[w])
                                    state["wide_done"] = ev
                                    held[0] = False
                                    wide.release()

                            if mode == 3:  # half-phase offset: the next batch may start voxelising once this one's network is enqueued
                                self.pipes[w].model_inference.on_network_done = release
                            else:
                                self.pipes[w].skeletonizer.on_wide_phase_done = release
                        try:
                            parts = self.pipes[w].process_clouds(clouds) if batches[i] > 1 else [self.pipes[w].process_cloud(cloud=clouds[0])]
                        finally:
                            if ordered:
                                release()  # (a batch without a single graph vertex never reaches the hook)
                                self.pipes[w].skeletonizer.on_wide_phase_done = None
                                self.pipes[w].model_inference.on_network_done = None
                        self.last = parts[-1]
                        if collect:
                            for k, sk in enumerate(parts):
                                finished.append(pack_skeleton(sk, cloud_id=self.rank * 1_000_000 + starts[i] + k))
                    self.streams[w].synchronize()
            except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
                state["error"] = e

        if S == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(w,)) for w in range(S)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if state["error"] is not None:
            raise state["error"]
        return finished

    def single_cloud(self, calls=SINGLE_CALLS):
        """SURVEY 8d's literal metric: `Pipeline.process_cloud(cloud=...)` on ONE cloud that sits in pinned host memory (the
        upload is part of the call, the skeleton is back on the host when it returns), one call at a time on one stream.
        Median wall time of `calls` calls on different clouds; then, with the kernel timers on, per-stage ms and the
        convolution family's roofline for a cloud alone on the GPU."""
        from smart_tree_amd import profiling
        from smart_tree_amd.data_types.cloud import Cloud

        pipe, st = self.pipes[0], self.streams[0]
        ids = [k % len(self.host) for k in range(calls)]

        def one(k):
            xyz, rgb = self.host[k]
            sk = pipe.process_cloud(cloud=Cloud(xyz=xyz.to(self.device, non_blocking=True), rgb=rgb.to(self.device, non_blocking=True)))
            st.synchronize()
            return sk

        with torch.cuda.stream(st):
            for k in ids[:2]:
                one(k)
            ms = []
            for k in ids:
                t0 = time.perf_counter()
                sk = one(k)
                ms.append(1e3 * (time.perf_counter() - t0))
            profiling.family_mode(False)
            profiling.enable(True)
            reps = min(3, len(ids))
            for k in ids[:reps]:
                one(k)
            torch.cuda.synchronize()
            profiling.enable(False)
            stage = profiling.stage_ms(reps)
            roof = profiling.roofline(HBM_PEAK_GBS, clouds_per_launch=1) or {}
        ms_sorted = sorted(ms)
        med = ms_sorted[len(ms_sorted) // 2]
        n_points = int(self.host[0][0].shape[0])
        gg = roof.get("gather_gemm") or {}
        return {"what": "Pipeline.process_cloud(cloud=...) with the cloud in pinned host memory: upload + full path + skeleton back on the "
                        "host, one call at a time (reference call shape: smart_tree/pipeline.py:55-93); median of the calls",
                "calls": len(ms), "ms": round(med, 3), "ms_min": round(ms_sorted[0], 3), "ms_max": round(ms_sorted[-1], 3),
                "points_per_s": n_points / (med * 1e-3), "stage_ms": stage,
                "roofline_gather_scatter": {k: gg.get(k) for k in ("hbm_frac", "achieved_GBps", "peak", "total_ms", "launches", "useful_TFLOPs")},
                "largest_kernel": {k: roof.get(k) for k in ("kernel", "frac", "avg_us", "launches", "total_ms")},
                "branches_last_call": int(sum(len(t.branches) for t in sk.skeletons))}


def extra_configs(device):
    """The other single-GPU configurations of BASELINE.json (untimed region, for the record), each as a BATCH through
    Pipeline.process_clouds with the kernel timers on: ms per cloud, the gather / rule-GEMM roofline entry of that batch, and
    what the skeleton stage did.  configs[3]: 5M-point dense canopy at 1 cm (2 clouds per launch set); configs[4]:
    peach-forest-65 in half-precision storage mode at 1M / 2 cm (8 clouds per launch set)."""
    from smart_tree_amd import profiling
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.synthetic import sample_tree_cloud

    out = {}
    for key, n, nb, kw, pk in (
            ("configs[4] peach-forest-65 fp16 storage, 1M pts, 2 cm", 1_000_000, 8, {}, dict(weights="peach-forest-65", fp16=True)),
            ("configs[3] dense canopy, 5M pts, 1 cm", 5_000_000, 2, dict(seed=3, foliage_fraction=0.6), dict(voxel=0.01)),
            ("configs[3] in the opt-in whole-cloud voxelisation mode (SURVEY 8f.2: no halo copies; NOT the reference's "
             "per-block results)", 5_000_000, 2, dict(seed=3, foliage_fraction=0.6), dict(voxel=0.01, blocking="whole"))):
        clouds = []
        for b in range(nb):
            c = sample_tree_cloud(n, seed=kw.get("seed", 0) + b, **{k: v for k, v in kw.items() if k != "seed"})
            clouds.append(Cloud(xyz=torch.from_numpy(c["xyz"]).to(device), rgb=torch.from_numpy(c["rgb"]).to(device)))
        pipe = build_pipeline(device, **pk)
        pipe.process_clouds(clouds)
        pipe.process_clouds(clouds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            sks = pipe.process_clouds(clouds)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / (reps * nb)
        profiling.enable(True)
        pipe.process_clouds(clouds)
        torch.cuda.synchronize()
        profiling.enable(False)
        roof = profiling.roofline(HBM_PEAK_GBS) or {}
        gg = roof.get("gather_gemm") or {}
        lc = pipe.last_labelled_cloud
        branches = int(sum(len(t.branches) for sk in sks for t in sk.skeletons))
        out[key] = {"ms_per_cloud": round(ms, 2), "points_per_s": round(n / ms * 1e3), "clouds_per_launch_set": nb,
                    "voxels_per_cloud": int((len(lc._base) if hasattr(lc, "_base") else len(lc)) / nb), "branches_per_cloud": branches / nb,
                    "skeleton_stage": "runs" if branches else "empty: this checkpoint's log-radius head is a constant (-6.24: 1.9 mm) on the "
                                                               "synthetic trees, so the outlier filter (8 neighbours within the own radius) "
                                                               "removes every point -- every voxel IS classed branch; the time is voxelise + "
                                                               "network + class filter + outlier filter",
                    "gather_gemm": {k: gg.get(k) for k in ("hbm_frac", "achieved_GBps", "useful_TFLOPs", "f32_matrix_frac", "total_ms", "launches")},
                    "stage_ms_per_cloud": profiling.stage_ms(nb)}
        del pipe, clouds
        torch.cuda.empty_cache()
    return out


def _skeleton_signature(trees):
    """(tree, branch id, parent id, xyz bytes, radii bytes) of every branch: equal signatures = identical skeletons."""
    sig = []
    for t, tree in enumerate(trees):
        for k, b in tree.branches.items():
            xyz = b.xyz.numpy() if hasattr(b.xyz, "numpy") else b.xyz
            rad = b.radii.numpy() if hasattr(b.radii, "numpy") else b.radii
            sig.append((t, int(k), int(b.parent_id), np.ascontiguousarray(xyz, np.float32).tobytes(),
                        np.ascontiguousarray(rad, np.float32).reshape(-1).tobytes()))
    return sig


def _select_phase_table(ticks) -> dict:
    """k_sk_select's in-kernel phase timers (tuning code 15; 100 MHz wall clock, summed over the components of the call)."""
    t = ticks.cpu().numpy()
    us = lambda i: round(float(t[i]) / 100.0, 1)
    return {"rounds": int(t[8]), "speculated_slots": int(t[13]), "commits": int(t[12]), "whole_workgroup_claims": int(t[9]),
            "claims_shared_with_helpers": int(t[14]), "candidates": int(t[11]),
            "phase_us": {"head": us(0), "window": us(7), "prune": us(1), "walk_rows": us(2), "claim": us(3), "replay_commit": us(4),
                         "one_mode": us(5), "long_mode": us(6)}}


def gt_branch_clouds(device, xyz_list, mv_list, voxel):
    """Branch clouds with the GENERATOR's exact medial vectors: every cloud centred (CentreCloud) and voxelised on the GPU exactly
    as ModelInference does, its inner representative points paired with the ground-truth medial vector of that point -- the regime
    the skeleton stage is built for (reference skeleton/skeletonize.py:33-47: medial points collapsed onto the branch axes)."""
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
    from smart_tree_amd.dataset.dataset import voxelize_blocks

    pre = AugmentationPipeline([CentreCloud()])
    out = []
    for xyz, mv in zip(xyz_list, mv_list):
        c = pre(Cloud(xyz=xyz.to(device), rgb=None))
        vb = voxelize_blocks(c.xyz, None, voxel)
        keep = vb.mask.bool().nonzero().view(-1)
        out.append(Cloud(xyz=vb.feats[keep, :3].contiguous(), medial_vector=mv.to(device)[vb.point_index[keep]].contiguous()))
    return out


def representative_inputs(device, host_xyz, host_mv, n_set=20, n_points=N_POINTS, calls=5, check_parity=True, training_scale=True):
    """The skeleton stage on inputs of the regime the reference runs it in (verdict of round 4: the shipped checkpoints' outputs
    on the 10 m benchmark tree at 2 cm are far outside their training distribution, so the skeleton stage of the headline consumes
    an arbitrary graph of a few hundred short branches):
      (a) configs[1]'s trees with their GROUND-TRUTH medial vectors -> Skeletonizer.forward + post_process: one cloud per call and
          `n_set` clouds in one launch set; ms, graph size, SSSP rounds, the selection's rounds / phases; the single-call result is
          compared with the oracle (identical = ids, parents, coordinates, radii of every branch);
      (b) a tree of the checkpoint's training scale (generator scale 0.3: ~3 m tall; 1 cm voxels, reference conf/training.yaml:18,
          46-47) through the WHOLE pipeline, compared with the oracle's skeleton of the GPU's labelled cloud, with the network's
          accuracy against the synthetic ground truth (direction cosine, log-radius) beside it."""
    from oracle import pipeline_oracle as po
    from smart_tree_amd import profiling
    from smart_tree_amd.data_types.cloud import Cloud
    from smart_tree_amd.dataset.dataset import voxelize_blocks
    from smart_tree_amd.skeleton import skeletonize, tuning
    from smart_tree_amd.synthetic import sample_tree_cloud

    out = {}
    pipe = build_pipeline(device)
    sk = pipe.skeletonizer
    n_set = min(n_set, host_xyz.shape[0], host_mv.shape[0])
    clouds = gt_branch_clouds(device, [host_xyz[j] for j in range(n_set)], [host_mv[j] for j in range(n_set)], VOXEL)

    def run(cloud):
        s = sk.forward(cloud)
        pipe.post_process(s)
        _ = s.skeletons  # materialise: device post-processing, one device-to-host copy
        parts = s.split()
        torch.cuda.synchronize()
        return s, parts

    def timed(cloud, reps):
        run(cloud)
        ms = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(cloud)
            ms.append(1e3 * (time.perf_counter() - t0))
        return sorted(ms)[len(ms) // 2]

    def profile(cloud, n_clouds):
        ticks = torch.zeros(32, dtype=torch.int64, device=device)
        skeletonize.reset_helper_stats()
        profiling.family_mode(False)
        profiling.enable(True)
        with tuning.override({tuning.TICKS: ticks.data_ptr()}):
            s, parts = run(cloud)
        profiling.enable(False)
        table = _select_phase_table(ticks)
        rows = profiling.kernel_table()
        sel = rows.get("k_sk_select")
        hs = skeletonize.helper_stats()
        return s, parts, {"stage_ms_per_cloud": profiling.stage_ms(n_clouds), "select": table,
                          "select_launch_ms": None if sel is None else round(sel["total_ms"], 3),
                          "select_algorithmic_bytes": None if sel is None else sel["bytes_per_launch"] * sel["launches"],
                          "helpers": hs, "sssp_rounds": int(skeletonize.last_run_stats.get("sssp_rounds", -1))}

    # (a) one cloud per call
    ms1 = timed(clouds[0], calls)
    s, parts, prof = profile(clouds[0], 1)
    entry = {"what": "Skeletonizer.forward + post_process on the seed-0 tree's inner voxel representatives with the generator's exact "
                     "medial vectors; median of %d calls, results on the host" % calls,
             "ms": round(ms1, 3), "graph_vertices": int(len(clouds[0])), "trees": len(s.skeletons),
             "branches": int(sum(len(t.branches) for t in s.skeletons)), **prof}
    if check_parity:
        c0 = clouds[0]
        trees = po.skeleton_from_labelled(c0.xyz.cpu().numpy(), c0.medial_vector.cpu().numpy(), np.zeros((len(c0), 1), np.float32))
        po.post_process(trees)
        entry["parity"] = "identical" if _skeleton_signature(s.skeletons) == _skeleton_signature(trees) else "DIFFERENT"
        entry["oracle_branches"] = int(sum(len(t.branches) for t in trees))
    out["configs[1] tree, ground-truth medial vectors, one cloud per call"] = entry
    # (a) n_set clouds in one launch set
    if n_set > 1:
        batch = Cloud.collate(clouds)
        msb = timed(batch, 3)
        s, parts, prof = profile(batch, n_set)
        same_first = _skeleton_signature(parts[0].skeletons) == _skeleton_signature(
            run(clouds[0])[0].skeletons)
        out["configs[1] trees (seeds 0..%d), ground-truth medial vectors, one launch set" % (n_set - 1)] = {
            "clouds": n_set, "ms_per_set": round(msb, 3), "ms_per_cloud": round(msb / n_set, 3), "graph_vertices": int(len(batch)),
            "branches": int(sum(len(t.branches) for p_ in parts for t in p_.skeletons)),
            "first_cloud_equals_its_single_call": bool(same_first), **prof}
    del clouds
    if not training_scale:
        return out
    # (b) a tree of the training scale through the whole pipeline
    c = sample_tree_cloud(n_points, seed=0, scale=0.3)
    pipe1 = build_pipeline(device, voxel=0.01)
    cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(device), rgb=torch.from_numpy(c["rgb"]).to(device))
    pipe1.process_cloud(cloud=cloud)
    ms = []
    for _ in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        skel = pipe1.process_cloud(cloud=cloud)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
    profiling.family_mode(False)
    profiling.enable(True)
    skel = pipe1.process_cloud(cloud=cloud)
    torch.cuda.synchronize()
    profiling.enable(False)
    stage = profiling.stage_ms(1)
    lc = pipe1.last_labelled_cloud
    xyz, mv, cls = lc.xyz.cpu().numpy(), lc.medial_vector.cpu().numpy(), lc.class_l.cpu().numpy()
    entry = {"what": "sample_tree_cloud(%d, seed 0, scale 0.3) -- a ~3 m tree, the size of the checkpoint's 4 m training crops -- at 1 cm "
                     "voxels through Pipeline.process_cloud (cloud resident in HBM); median of %d calls" % (n_points, calls),
             "ms": round(sorted(ms)[len(ms) // 2], 3), "voxels_inner": int(len(lc)),
             "branch_voxels": int((cls.reshape(-1) == 0).sum()), "trees": len(skel.skeletons),
             "branches": int(sum(len(t.branches) for t in skel.skeletons)), "stage_ms": stage}
    # the network against the synthetic ground truth of the same representative points
    pre = pipe1.preprocessing(cloud)
    vb = voxelize_blocks(pre.xyz, None, 0.01)
    keep = vb.mask.bool().nonzero().view(-1)
    if int(keep.shape[0]) == len(lc):
        gt = torch.from_numpy(c["medial_vector"]).to(device)[vb.point_index[keep]].cpu().numpy().astype(np.float64)
        r_gt, r_net = np.linalg.norm(gt, axis=1), np.linalg.norm(mv.astype(np.float64), axis=1)
        ok = (r_gt > 0) & (r_net > 0) & np.isfinite(r_net)
        cos = np.sum(gt[ok] * mv[ok], axis=1) / (r_gt[ok] * r_net[ok])
        lr_gt, lr_net = np.log(r_gt[ok]), np.log(r_net[ok])
        entry["network_vs_ground_truth"] = {
            "direction_cosine_mean": float(np.mean(cos)), "direction_cosine_median": float(np.median(cos)),
            "log_radius_median_net": float(np.median(lr_net)), "log_radius_median_truth": float(np.median(lr_gt)),
            "log_radius_correlation": float(np.corrcoef(lr_gt, lr_net)[0, 1]) if ok.sum() > 2 else None,
            "log_radius_abs_err_median": float(np.median(np.abs(lr_net - lr_gt))),
            "classed_branch_fraction": float((cls.reshape(-1) == 0).mean())}
    if check_parity:
        trees = po.skeleton_from_labelled(xyz, mv, cls)
        po.post_process(trees)
        entry["parity"] = "identical" if _skeleton_signature(skel.skeletons) == _skeleton_signature(trees) else "DIFFERENT"
    out["training-scale tree (scale 0.3, 1 cm), whole pipeline"] = entry
    return out


def representative_e2e(worker, host_mv, steps, B, points, passes=3):
    """ONE end-to-end figure on the regime the skeleton stage is written for (round-5 verdict): the same `steps` clouds through
    the WHOLE path -- upload excluded, CentreCloud, blocks / voxels, the network (it runs and is timed), class filter -- but the
    skeleton stage consumes the GENERATOR's exact medial vector of every voxel's representative point instead of the network's
    output (the shipped checkpoints' output on the 10 m benchmark tree carries no information, DESIGN.md section 4: the headline's
    skeleton stage works on an arbitrary graph that is 2-3x cheaper than this one).  Launch sets as in the timed region
    (`plan_batches`, one set at a time on one stream); median of `passes` passes; the first cloud's skeleton is compared with the
    oracle's skeleton of the same labelled points (identical = ids, parents, coordinates, radii of every branch)."""
    from oracle import pipeline_oracle as po
    from smart_tree_amd import profiling
    from smart_tree_amd.data_types.cloud import Cloud, MaskedCloud

    dev, pipe, st = worker.device, worker.pipes[0], worker.streams[0]
    plan = plan_batches(steps, 1, B)
    starts = np.concatenate([[0], np.cumsum(plan)]).tolist()
    n_host = host_mv.shape[0]
    with torch.cuda.stream(st):
        gts = [torch.cat([host_mv[(a + k) % n_host].to(dev) for k in range(sz)]) for a, sz in zip(starts, plan)]  # resident, like the inputs

        def one_set(i):
            clouds = worker.batch_clouds(starts[i], plan[i])
            batch = Cloud.collate([Cloud(c.xyz, c.rgb) for c in clouds]) if plan[i] > 1 else clouds[0]
            with profiling.stage("preprocess"):
                batch = pipe.preprocessing(batch)
            base, mask = pipe.model_inference.forward(batch).pending()
            with profiling.stage("class_filter"):
                lc = Cloud(xyz=base.xyz, rgb=base.rgb, medial_vector=gts[i].index_select(0, pipe.model_inference.last_point_index),
                           class_l=torch.zeros_like(base.class_l), seg_off=base.seg_off)
                branch = MaskedCloud(lc, mask).filter_by_class(pipe.branch_classes)
            sk = pipe.skeletonizer.forward(branch)
            with profiling.stage("post_process"):
                pipe.post_process(sk)
                parts = sk.split()
            return parts, lc, mask

        def one_pass():
            first_set = None
            for i in range(len(plan)):
                r = one_set(i)
                first_set = r if i == 0 else first_set
            st.synchronize()
            return first_set

        one_pass()
        ms = []
        for _ in range(passes):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            first = one_pass()
            ms.append(1e3 * (time.perf_counter() - t0))
        profiling.family_mode(True)
        profiling.enable(True)
        one_pass()
        torch.cuda.synchronize()
        profiling.enable(False)
        stage = profiling.stage_ms(steps)
        parts, lc, mask = first
        a, b = (0, len(lc)) if lc.seg_off is None else [int(v) for v in lc.seg_off[:2].tolist()]
        keep = mask[a:b].nonzero().view(-1) + a
        xyz, mv = lc.xyz[keep].cpu().numpy(), lc.medial_vector[keep].cpu().numpy()
    trees = po.skeleton_from_labelled(xyz, mv, np.zeros((len(xyz), 1), np.float32))
    po.post_process(trees)
    med = sorted(ms)[len(ms) // 2]
    return {"what": "the timed region's %d clouds through the whole path with the skeleton stage fed the generator's exact medial vectors "
                    "(network still run and timed; inputs resident; launch sets %s, one at a time); median of %d passes" % (steps, plan, passes),
            "value": steps * points / (med * 1e-3), "unit": "points/s", "ms_per_step": med / steps,
            "passes_ms_per_step": [round(v / steps, 4) for v in ms], "stage_ms_per_cloud": stage,
            "graph_vertices_first_cloud": int(len(xyz)), "branches_first_cloud": int(sum(len(t.branches) for t in parts[0].skeletons)),
            "parity": "identical" if _skeleton_signature(parts[0].skeletons) == _skeleton_signature(trees) else "DIFFERENT"}


def strong_scaling_projection(worker, S, B, points, reps=3):
    """BASELINE.json configs[2] is a FIXED batch of 64 clouds over 8 GPUs (8 per GPU).  No 8-GPU node is the builder's to use, so
    this is a PROJECTION from one GPU, NOT a scaling measurement: t(n) = one launch set of n of the 64 clouds (seeds 0..63) alone
    on the GPU, inputs resident, median of `reps`; t64 = all 64 on this one GPU with the timed region's own plan.  With the clouds
    dealt round-robin, every rank of 8 holds 8 clouds and the job lasts as long as its slowest rank: the 8 sets of 8 are timed
    one after the other here and the slowest stands for the 8-GPU job (the gather of KBs per cloud is not in it)."""
    def timed(plan_, streams, first=0):
        starts_backup = worker.first
        worker.first = first
        try:
            worker.run(plan_, collect=False, streams=streams)
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                worker.run(plan_, collect=False, streams=streams)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
        finally:
            worker.first = starts_backup
        return sorted(ts)[len(ts) // 2]

    n_all = PROJECTION_CLOUDS
    sets = {n: round(timed([n], 1), 3) for n in (8, 16, 32, 64)}
    plan64 = plan_batches(n_all, S, B)
    t64 = min(timed(plan64, S), sets[64])
    rank_ms = [round(timed([8], 1, first=8 * r), 3) for r in range(8)]  # rank r of 8: a contiguous eighth stands for its round-robin share
    slowest = max(rank_ms)
    return {"note": "PROJECTION from one GPU, NOT a scaling measurement (no 8-GPU node available to the builder): configs[2] = 64 clouds "
                    "split over 8 ranks = 8 clouds per rank; speed-up = t(64 clouds on this GPU) / t(slowest set of 8)",
            "ms_one_launch_set": sets, "ms_64_clouds_one_gpu": round(t64, 3), "plan_64_clouds": plan64,
            "ms_sets_of_8": rank_ms, "projected_speedup_8gpu": round(t64 / slowest, 2),
            "projected_value_8gpu": n_all * points / (slowest * 1e-3),
            "target": ">= 6.5x (BASELINE.json north_star)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=384)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4] batches")
    ap.add_argument("--free-running", action="store_true", help="one GPU: an extra untimed pass with 3 batches in flight and no "
                    "ordering of their chip-filling phases, reported as `free_running` (profiles/r02_free_running.txt)")
    ap.add_argument("--batch", type=int, default=MAX_BATCH, help="clouds per launch set (Pipeline.process_clouds); 1 = one cloud per call")
    ap.add_argument("--streams", type=int, default=STREAMS, help="batches in flight per GPU (one host thread + HIP stream each)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): every rank runs --steps clouds of its own.  strong (BASELINE.json configs[2]): "
                         "a FIXED batch of --clouds independent clouds (seeds 0 .. clouds-1) is split round-robin over the ranks, `value` = "
                         "clouds x points / the slowest rank's time, the skeletons gathered to rank 0 inside the timed region")
    ap.add_argument("--clouds", type=int, default=0, help="--scaling strong: clouds of the whole job (default: --steps; configs[2]: 64)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    from smart_tree_amd.sharding import shard_indices

    strong = args.scaling == "strong"
    total_clouds = (args.clouds or args.steps) if strong else world * args.steps
    # K = the steps THIS rank runs per pass: --steps (weak) or its round-robin share of the fixed batch (strong)
    my_clouds = shard_indices(total_clouds, rank, world) if strong else None
    K = len(my_clouds) if strong else args.steps
    assert K > 0, f"--scaling strong: {total_clouds} clouds cannot feed {world} ranks"
    # the synthetic clouds of this rank, generated by forked processes BEFORE this process has a GPU context: every step of a
    # pass is a different cloud (BASELINE configs[2]: seeds 0..63; rank r draws r * D .. r * D + D - 1, rank 0's first is seed 0)
    n_distinct = max(1, min(MAX_DISTINCT, K))
    t_gen = time.perf_counter()
    want_rep = world == 1 and not args.no_extras
    # one GPU: configs[2]'s 64 seeds are generated whatever --steps says (the timed region uses the first n_distinct of them; the
    # strong-scaling projection times launch sets of 8 .. 64 clouds), each with the generator's exact medial vectors (representative_e2e)
    n_gen = max(n_distinct, PROJECTION_CLOUDS) if want_rep and not strong else n_distinct
    n_rep = n_gen if want_rep else 0
    seeds = ([my_clouds[j % K] for j in range(n_gen)] if strong else [rank * n_distinct + j for j in range(n_gen)])
    gen = generate_clouds(args.points, seeds, max(1, usable_cores() // max(world, 1)), n_mv=n_rep)
    host_xyz, host_mv = gen if n_rep > 0 else (gen, None)
    t_gen = time.perf_counter() - t_gen
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # ST_BENCH_DRYRUN=1 (developer aid): exercise the multi-rank control flow on a box with ONE GPU -- every rank uses
    # cuda:0 and the result gather runs over gloo with host tensors.  Never set by the driver.
    dryrun = os.environ.get("ST_BENCH_DRYRUN") == "1"
    if dryrun:
        local_rank = 0
    # host waits: spinning (the HIP default) when the host has cores to spare -- a worker thread then keeps a core busy, and a
    # read-back returns a few tens of microseconds earlier (20 steps: 1.57-1.62 against 1.63-1.66 ms per step) --, blocking
    # otherwise (eight ranks with two worker threads each on a small host)
    blocking = os.environ.get("ST_BENCH_BLOCKING_SYNC", "0" if usable_cores() >= 8 * max(world, 1) else "1") == "1"
    if blocking:
        # host waits block on the completion interrupt instead of spinning (hipDeviceScheduleBlockingSync = 4; must be set
        # before the device's context exists).  Measured: 0.2 host cores busy instead of 1.93 for two batches in flight, at
        # 1.297-1.302 instead of 1.288-1.292 ms per cloud -- eight ranks need two cores, not sixteen.
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipSetDevice(local_rank) == 0 and hip.hipSetDeviceFlags(4) == 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = torch.device("cpu") if dryrun else device
    global ORDERED
    if "ST_BENCH_ORDERED" not in os.environ and K < SHORT_RUN_STEPS:
        ORDERED = 0  # measured at the driver's --steps 20: 1.83 ms per step free-running against 1.96 in turns (profiles/r03_sweep_plan.txt)

    from smart_tree_amd import profiling
    from smart_tree_amd.sharding import gather_skeletons

    # host side of the GPU phase: a few Python threads issuing launches; torch's CPU ops here are tiny (offset lists), and an
    # OpenMP pool spinning on every core beside them only takes cycles from the launch threads (the CPU baseline sets its own)
    torch.set_num_threads(int(os.environ.get("ST_BENCH_TORCH_THREADS", "1")))

    # host budget: with spinning waits a worker thread keeps one core busy (1.92 cores for 2 batches in flight, 1.01 for one) and
    # the HIP runtime's helper threads, the main thread and the collective's proxy want their share -- two cores per worker thread
    # and one per rank, or fewer batches in flight (one in flight costs 14 % at 64 clouds per batch: 1.47 instead of 1.29 ms per
    # cloud; a throttled cgroup costs more).  With blocking waits (the default) a worker needs a fifth of a core.
    S = max(1, min(args.streams, max(1, (usable_cores() - world) // ((1 if blocking else 2) * max(world, 1)))))
    S = max(1, min(S, K // 16), min(S, 3, K // 6))  # see plan_batches
    B = max(1, min(args.batch, 64))
    finished = []  # packed skeletons of this rank, gathered to rank 0 once per timed region (no per-step rendezvous:
    #                 clouds differ in cost, a collective per step would make every step as slow as its slowest rank)
    # for the record (one GPU only): the same K steps free-running -- the chip-filling phases of the batches in flight overlap,
    # which fills the bubbles at their host round trips (a few % more throughput) and stretches every kernel's launch bracket
    S_free = max(1, min(FREE_STREAMS, usable_cores(), max(1, K // 16))) if world == 1 and ORDERED in (1, 3) and args.free_running else 0
    # (one GPU: the strong-scaling projection runs 64 clouds with the plan the default command would use -- its worker threads exist)
    S_proj = max(1, min(args.streams, max(1, (usable_cores() - 1) // (1 if blocking else 2)), PROJECTION_CLOUDS // 16)) if want_rep and not strong else 0
    worker = CloudWorker(device, max(S, S_free, S_proj), host_xyz, rank)

    def run_steps(total, upload=False):
        plan_ = plan_batches(total, S, B)
        if upload and len(plan_) == 1 and total >= 8 and S > 1:
            # with the uploads inside the pass, a lone batch would copy all of its clouds before its first kernel: two launch sets on
            # two streams instead, a small one whose copies are exposed (6 of 20 clouds) and a large one whose copies run under the
            # small one's kernels.  Measured (profiles/r04_sweep_upload_plans.txt, 20 steps): 10 + 10 2.08-2.13 ms per step,
            # 6 + 14 1.80-1.84, 4 + 16 1.84-1.86, 8 + 12 1.83-1.86, 2 + 18 1.84-1.88, 3 + 5 + 12 1.87-1.95.
            k = max(2, int(round(0.3 * total)))
            plan_ = [k, total - k]
        finished.extend(worker.run(plan_, collect=world > 1, upload=upload, streams=S))

    def gather():
        if world > 1:
            gather_skeletons(finished, device=coll_device)
            finished.clear()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()  # inputs and weights are resident before any worker stream touches them
    # warm-up: at least --warmup clouds, and one untimed pass over the very batches the timed region will run.  Every stream
    # has its own allocator pool, and batches differ a little in their voxel / vertex counts: a pool that has seen only one
    # batch still grows during the next few (hipMalloc synchronises the device: measured 2.1 instead of 1.55 ms per cloud
    # when the timed region started after one warm-up batch per stream)
    plan = plan_batches(K, S, B)
    # SETTLE (untimed, reported as `settle_s` / `settle_steps`; not the warm-up the flag asks for): whole passes over the very batches
    # the timed region will run, until two consecutive passes agree to 5 % AND the process is MIN_UPTIME seconds old.  Why: every
    # stream has its own allocator pool, and batches differ a little in their voxel / vertex counts -- a pool that has seen only
    # one batch still grows during the next few (hipMalloc synchronises the device: measured 2.1 instead of 1.55 ms per cloud when
    # the timed region started after one warm-up batch per stream); and a process that starts right after another GPU process has
    # exited -- the round-end sequence pytest -> smoke -> bench -- runs with inflated host round trips for its first ~15-20 s on the
    # gpurun boxes (profiles/r02_after_another_process.txt: passes agree with each other at 2.5-2.9 ms per cloud, then drop to 1.3).
    settle_steps, t_settle = 0, time.perf_counter()
    warm_last_ms = None
    if plan and MIN_UPTIME >= 0:
        prev = None
        while True:
            finished.clear()  # (multi-rank: only the last pass is gathered)
            profiling.family_mode(True)
            profiling.enable(True)  # the settle passes run exactly what the timed pass runs, kernel timers included
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            finished.extend(worker.run(plan, collect=world > 1, streams=S))
            torch.cuda.synchronize()
            cur = time.perf_counter() - t0
            settle_steps += sum(plan)
            uptime = time.perf_counter() - T_PROCESS
            if (prev is not None and abs(cur - prev) <= 0.05 * prev and uptime >= MIN_UPTIME) or uptime > MIN_UPTIME + 45.0:
                break
            prev = cur
        profiling.enable(False)
        warm_last_ms = 1e3 * cur / max(sum(plan), 1)
    settle_s = time.perf_counter() - t_settle
    # WARM-UP: exactly --warmup steps, untimed, right before the timed region
    warm = max(0, int(args.warmup))
    if warm:
        finished.clear()
        finished.extend(worker.run(plan_batches(warm, S, B), collect=world > 1, streams=S))
    gather()
    fence()
    single = worker.single_cloud() if world == 1 and args.warmup > 0 else None  # (--warmup 0: the PMC passes of tools/collect_r5.sh want the batch's launches alone)
    fence()
    # kernel timers of the timed region: the convolutions of a forward pass are bracketed ONCE as a family (a pair of events
    # around each of the 26 launches cost ~10 us apiece between kernels that otherwise run back to back); the per-class table
    # comes from the solo pass below.  The timers stay on over all passes: launch durations are averages over them.
    profiling.family_mode(True)
    profiling.enable(True)
    import gc
    gc.collect()
    gc.disable()  # a short run is a pass of ~30 ms: a collection of the launch threads' garbage inside it is a millisecond
    pass_s, host_cores = [], []
    try:
        for _ in range(PASSES):
            fence()
            cpu0 = os.times()
            t0 = time.perf_counter()
            run_steps(K)  # returns when every worker has synchronised its streams
            gather()  # inside the timed region: the skeletons of all ranks end up on rank 0
            fence()
            dt_pass = time.perf_counter() - t0
            cpu1 = os.times()
            pass_s.append(dt_pass)
            profiling.settle()  # (between the timed brackets: the byte counts of this pass, so that its tables can be freed)
            host_cores.append(((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(dt_pass, 1e-9))  # this rank's process
    finally:
        gc.enable()
    profiling.enable(False)
    roof = profiling.roofline(HBM_PEAK_GBS, clouds_per_launch=max(plan_batches(K, S, B)))
    stage_ms = profiling.stage_ms(K * PASSES)
    sk = worker.last
    last = {"trees": len(sk.skeletons), "branches": int(sum(len(t.branches) for t in sk.skeletons))}
    # the same K clouds with the host -> device upload of every cloud inside the pass (pinned host buffers, copy streams)
    up_s = []
    run_steps(K, upload=True)  # (untimed: the copy streams' allocator pools)
    gather()
    for _ in range(UPLOAD_PASSES):
        fence()
        t1 = time.perf_counter()
        run_steps(K, upload=True)
        gather()
        fence()
        up_s.append(time.perf_counter() - t1)
    free = None
    if S_free > 1:
        plan_free = plan_batches(K, S_free, B)
        worker.run(plan_free, collect=False, streams=S_free, mode=0)  # untimed: the third stream's allocator pool
        torch.cuda.synchronize()
        profiling.enable(True)
        t2 = time.perf_counter()
        worker.run(plan_free, collect=False, streams=S_free, mode=0)
        torch.cuda.synchronize()
        dt_free = time.perf_counter() - t2
        profiling.enable(False)
        rf = profiling.roofline(HBM_PEAK_GBS, clouds_per_launch=max(plan_free))
        free = {"note": "the same %d steps with %d batches in flight and NO ordering of their chip-filling phases (untimed extra "
                        "pass): kernels of different batches share the chip, brackets stretch" % (K, S_free),
                "value": K * args.points / dt_free, "ms_per_step": 1e3 * dt_free / K,
                "batches_in_flight": S_free, "roofline_kernel": rf["kernel"], "roofline_frac": rf["frac"],
                "gather_gemm_hbm_frac": (rf.get("gather_gemm") or {}).get("hbm_frac")}
    # for the record: ONE batch at a time on one stream with the kernel timers on -- solo kernel durations (in the timed region
    # the kernels of the batches in flight share the chip, which inflates every bracket)
    roof_solo = None
    if world == 1:
        profiling.family_mode(False)  # every launch bracketed: the per-class table
        profiling.enable(True)
        worker.run([min(B, max(K, 1))], collect=False, streams=1)
        torch.cuda.synchronize()
        profiling.enable(False)
        full = profiling.roofline(HBM_PEAK_GBS, clouds_per_launch=min(B, max(K, 1)))  # (PMC traffic scaled to this batch)
        roof_solo = {"note": "one batch of %d clouds alone on the GPU (untimed extra pass): solo launch durations" % min(B, max(K, 1))}
        roof_solo.update({k: full.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launches", "avg_us",
                                                  "algorithmic_bytes_per_launch", "branch_selection", "gather_gemm", "all_kernels")})
    rep_e2e = projection = None
    if want_rep and not strong and host_mv is not None:
        rep_e2e = representative_e2e(worker, host_mv, K, B, args.points)
        projection = strong_scaling_projection(worker, S_proj, B, args.points)
    per_rank = None
    if world > 1:  # every pass: the slowest rank
        mine = torch.tensor([sorted(pass_s)[len(pass_s) // 2], sorted(host_cores)[len(host_cores) // 2]], dtype=torch.float64, device=coll_device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # (outside the timed region: each rank's own median pass and busy host cores, for the record)
        per_rank = {"median_pass_ms": [round(1e3 * float(e[0]), 2) for e in every], "host_cpu_cores_busy": [round(float(e[1]), 2) for e in every]}
        t = torch.tensor(pass_s + up_s, dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pass_s, up_s = t[:PASSES].tolist(), t[PASSES:].tolist()
    med = lambda v: sorted(v)[len(v) // 2]
    dt, dt_up = med(pass_s), med(up_s)

    if rank == 0:
        value = total_clouds * args.points / dt
        batches = plan_batches(K, S, B)
        gg = (roof or {}).pop("gather_gemm", None)
        out = {
            "metric": "points/sec end-to-end (voxelize->sparse-UNet->skeleton), 1M-pt tree",
            "value": value, "unit": "points/s", "n_gpus": world, "steps": total_clouds if strong else K, "warmup": warm,
            "ms_per_step": 1e3 * dt / (total_clouds if strong else K), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "passes": {"n": PASSES, "statistic": "median", "ms_per_step": [round(1e3 * v / (total_clouds if strong else K), 4) for v in pass_s],
                       "value_min": total_clouds * args.points / max(pass_s), "value_max": total_clouds * args.points / min(pass_s)},
            "value_incl_host_upload": total_clouds * args.points / dt_up,
            "incl_host_upload": {"passes": UPLOAD_PASSES, "statistic": "median", "ms_per_step": [round(1e3 * v / (total_clouds if strong else K), 4) for v in up_s],
                                 "note": "every cloud of the pass is copied from pinned host memory inside the pass (24 bytes per point: xyz + rgb); "
                                         "a worker enqueues its NEXT batch's copies on a copy stream before the current batch's kernels"},
            "config": {"workload": f"configs[1]: one {args.points}-point synthetic tree per rank per step, 2 cm voxels, "
                                   "noble-elevator-58 weights, full Pipeline path with prune/repair/smooth; steps run in "
                                   "batches of clouds through ONE launch set (Pipeline.process_clouds)",
                       "inputs": "resident in HBM when the timed region starts (the task's bench contract: the PCIe-inclusive rate is never `value`); "
                                 "SURVEY 8d's reading -- the cloud starts in (pinned) HOST memory -- is `value_incl_host_upload` for the same K steps and "
                                 "`single_cloud` for one Pipeline.process_cloud call at a time",
                       "distinct_clouds_per_rank": n_distinct,
                       "seeds": (f"strong scaling: the job's clouds are seeds 0..{total_clouds - 1}, rank r holds r, r + {world}, ..." if strong else
                                 f"{rank * n_distinct}..{rank * n_distinct + n_distinct - 1} (rank r: r * {n_distinct} + j)"),
                       "parallelism": f"cloud-sharded x{world}" + (f" (strong: {total_clouds} clouds in all, {K} on rank 0)" if strong else ""),
                       "strong_scaling_projection": projection,
                       "clouds_per_launch_set": max(batches), "batches_in_timed_region": len(batches),
                       "batches_in_flight_per_gpu": S, "host_threads_per_gpu": S,
                       "settle_s": round(settle_s, 1), "settle_steps": settle_steps,
                       "settle_note": "untimed passes over the timed region's own batches before the --warmup steps, until two passes agree "
                                      "to 5 % and the process is `warmup_until_process_age_s` old (allocator pools, clock ramp after "
                                      "another GPU process)",
                       "value_incl_host_upload": total_clouds * args.points / dt_up,
                       "ms_per_step_incl_host_upload": 1e3 * dt_up / (total_clouds if strong else K),
                       "single_cloud_ms": None if single is None else single["ms"],
                       "schedule": {1: "the chip-filling phases (voxelise .. adjacency) of the batches in flight take turns; a batch's "
                                       "skeleton stage (one compute unit per tree) overlaps with the other batch's chip-filling phase",
                                    3: "the batches in flight take turns with voxelise .. network (half a phase apart): a batch's searches / "
                                       "components / adjacency and its skeleton stage (one compute unit per tree) overlap with the other "
                                       "batch's voxelisation and network, two networks never share the chip",
                                    0: "free-running", 2: "conv sequences take turns"}.get(ORDERED if S > 1 else 0),
                       "warmup_until_process_age_s": MIN_UPTIME,
                       "cloud_generation_s": round(t_gen, 1),
                       "host_cpu_cores_busy_in_timed_region": round(med(host_cores), 2),
                       "host_waits": "blocking (hipDeviceScheduleBlockingSync)" if blocking else "spinning (HIP default)",
                       "host_cores_usable": usable_cores(), "per_rank": per_rank,
                       "dry_run": "ST_BENCH_DRYRUN=1: every rank on cuda:0, collectives over gloo -- multi-rank control flow and host "
                                  "contention only, NOT a scaling measurement" if dryrun else None,
                       "last_warmup_pass_ms_per_step": None if warm_last_ms is None else round(warm_last_ms, 3)},
            "parity_note": "results are checked against oracle/ (a CPU restatement pinned by goldens that the reference's own "
                           "glue code produced); the semantics of the reference's un-vendored third-party packages (spconv "
                           "voxel drop rule / hash order, FRNN tie order, cugraph tie-breaks) are restated, not pinned",
            "roofline": roof,
            "roofline_gather_scatter": gg,
            "single_cloud": single,
            "representative_e2e": rep_e2e,
            "roofline_solo": roof_solo,
            "free_running": free,
            "stage_ms": stage_ms,
            "last_result": last,
        }
        if world == 1 and not args.no_cpu_baseline:
            entry, cpu_lc = cpu_baseline(args.points)
            out["cpu_baseline"] = entry
            out.update(parity_in_run(worker.pipes[0], worker.clouds[0], cpu_lc))
        if world == 1 and not args.no_extras:
            from smart_tree_amd.skeleton import skeletonize
            out["helper_workgroups"] = skeletonize.helper_stats()  # lost > 0: a fall-back of the helper protocol ran (slower, not wrong)
            del worker
            torch.cuda.empty_cache()
            out["representative_inputs"] = representative_inputs(device, host_xyz, host_mv, n_set=min(REP_SET, n_rep), n_points=args.points)
            out["other_configs"] = extra_configs(device)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
