"""Host logic of bench.py that needs no GPU: launch-set plans and the strong-scaling split of BASELINE.json configs[2]."""
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("steps", [1, 5, 8, 20, 32, 33, 64, 100, 384])
@pytest.mark.parametrize("streams", [1, 2, 3])
def test_plan_covers_every_step_once(bench, steps, streams):
    plan = bench.plan_batches(steps, streams, bench.MAX_BATCH)
    assert sum(plan) == steps and all(1 <= b <= bench.MAX_BATCH for b in plan)
    if steps <= 32:
        assert plan == [steps]  # a short run is ONE launch set


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_strong_scaling_split_of_configs2(world):
    """--scaling strong --clouds 64: the fixed batch of 64 seeds is dealt round-robin; every seed runs on exactly one rank and the
    ranks' shares differ by at most one cloud (configs[2]: 8 clouds on each of 8 GPUs)."""
    from smart_tree_amd.sharding import shard_indices

    shares = [shard_indices(64, r, world) for r in range(world)]
    assert sorted(s for sh in shares for s in sh) == list(range(64))
    assert max(map(len, shares)) - min(map(len, shares)) <= 1
    assert all(len(sh) == 64 // world for sh in shares)
