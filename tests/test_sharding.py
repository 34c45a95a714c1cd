"""Cloud sharding + skeleton gather with a world_size-2 gloo process group on CPU."""
import importlib.util
import os
from pathlib import Path
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smart_tree_amd.data_types.branch import BranchSkeleton
from smart_tree_amd.data_types.tree import DisjointTreeSkeleton, TreeSkeleton
from smart_tree_amd.sharding import gather_skeletons, pack_skeleton, shard_indices, unpack_skeletons


def _fake_skeleton(seed: int) -> DisjointTreeSkeleton:
    g = torch.Generator().manual_seed(seed)
    trees = []
    for t in range(1 + seed % 2):
        branches = {}
        for b in range(2 + seed % 3):
            m = 2 + int(torch.randint(0, 5, (1,), generator=g))
            branches[b] = BranchSkeleton(b, b - 1, torch.rand((m, 3), generator=g), torch.rand((m, 1), generator=g))
        trees.append(TreeSkeleton(t, branches))
    return DisjointTreeSkeleton(trees)


def _same(a: DisjointTreeSkeleton, b: DisjointTreeSkeleton):
    assert len(a.skeletons) == len(b.skeletons)
    for ta, tb in zip(a.skeletons, b.skeletons):
        assert list(ta.branches) == list(tb.branches)
        for k in ta.branches:
            x, y = ta.branches[k], tb.branches[k]
            assert x.parent_id == y.parent_id and torch.equal(x.xyz, y.xyz) and torch.equal(x.radii, y.radii)


def test_shard_indices_partition():
    owned = [shard_indices(11, r, 4) for r in range(4)]
    assert sorted(i for o in owned for i in o) == list(range(11))
    assert owned[1] == [1, 5, 9]


def test_pack_roundtrip_single_process():
    sk = _fake_skeleton(3)
    tables, geoms = gather_skeletons([pack_skeleton(sk, cloud_id=7)])
    _same(unpack_skeletons(tables[0], geoms[0])[7], sk)
    empty = DisjointTreeSkeleton([])
    t, g = pack_skeleton(empty)
    assert t.shape == (0, 6) and g.shape == (0, 4) and unpack_skeletons(t, g) == {}


def _worker(rank: int, world: int, port: int, n_clouds: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_clouds, rank, world)
    tables, geoms = gather_skeletons([pack_skeleton(_fake_skeleton(i), cloud_id=i) for i in mine])
    if rank == 0:
        merged = {}
        for t, g in zip(tables, geoms):
            merged.update(unpack_skeletons(t, g))
        # numpy payloads are pickled by value (torch tensors would travel as shared-memory handles that die with the worker)
        q.put({k: tuple(a.numpy() for a in pack_skeleton(v, k)) for k, v in merged.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_clouds = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clouds, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == list(range(n_clouds))
    for i in range(n_clouds):
        t, g = (torch.from_numpy(a) for a in got[i])
        _same(unpack_skeletons(t, g)[i], _fake_skeleton(i))


@pytest.mark.gpu
def test_rccl_one_rank_group_runs_the_collectives():
    """No 8-GPU node is available to the builder: the "nccl" (= RCCL) branch at least EXECUTES here -- communicator init on
    cuda:0, the size all_gather + payload gathers of gather_skeletons on device tensors, and bench.py's all_reduce(MAX) and
    barrier -- on a one-rank group.  (Scaling itself stays unmeasured until the driver's SCALE run exists.)"""
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        packed = [pack_skeleton(_fake_skeleton(i), cloud_id=i) for i in range(3)]
        tables, geoms = gather_skeletons(packed, device=dev, always_collective=True)
        assert len(tables) == 1 and not tables[0].is_cuda
        merged = unpack_skeletons(tables[0], geoms[0])
        for i in range(3):
            _same(merged[i], _fake_skeleton(i))
        t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert t.tolist() == [1.5, 2.5]
    finally:
        dist.destroy_process_group()


def test_bench_batch_plan():
    """bench.py deals its K clouds out as batches (clouds per launch set): every cloud exactly once, no batch above the cap,
    a multiple of the stream count where K allows it (no ragged tail), never an empty batch."""
    spec = importlib.util.spec_from_file_location("bench_mod", Path(__file__).resolve().parents[1] / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for steps in (0, 1, 2, 5, 20, 47, 48, 96, 1000):
        for streams in (1, 2, 3, 4):
            for cap in (1, 8, 16):
                plan = bench.plan_batches(steps, streams, cap)
                assert sum(plan) == steps and all(0 < b <= cap for b in plan)
                if plan:
                    eff = max(1, min(streams, steps // 16), min(streams, 3, steps // 6))  # fewer batches in flight for short runs
                    assert max(plan) - min(plan) <= 1 and (len(plan) % eff == 0 or len(plan) == steps or plan == [steps])
    # a short run that fits one launch set is ONE batch (round 4); above the cap it is split evenly over the streams
    assert bench.plan_batches(20, 3, 16) == [7, 7, 6] and bench.plan_batches(96, 3, 16) == [16] * 6
    assert bench.plan_batches(20, 4, 24) == [20] and bench.plan_batches(192, 4, 24) == [24] * 8 and bench.plan_batches(5, 4, 24) == [5]
    assert bench.plan_batches(384, 3, 48) == [43] * 6 + [42] * 3 and bench.plan_batches(384, 3, 64) == [64] * 6
