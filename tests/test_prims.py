"""Device-wide scan / radix sort primitives against numpy."""
import numpy as np
import pytest
import torch

from smart_tree_amd import _lib


def _check_scan(backend, n):
    L = _lib.lib()
    a = np.random.RandomState(n).randint(0, 7, n).astype(np.int32)
    t = torch.from_numpy(a).to(backend)
    out = torch.zeros(max(n, 1), dtype=torch.int32, device=backend)
    total = torch.zeros(1, dtype=torch.int32, device=backend)
    ws = _lib.workspace(L.st_scan_workspace_bytes(n), backend)
    _lib.check(L.st_scan_u32(_lib.ptr(t), _lib.ptr(out), n, _lib.ptr(total), _lib.ptr(ws), ws.numel(), _lib.stream(backend)))
    ref = np.cumsum(a) - a
    np.testing.assert_array_equal(out.cpu().numpy()[:n], ref)
    assert int(total.cpu()[0]) == int(a.sum())


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 4095, 4096, 4097, 70001])
def test_exclusive_scan(backend, n):
    _check_scan(backend, n)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4_300_003,    # > 1024 tiles: the recursive tile-offset scan
                               51_000_001])  # the cell table of a batched kNN grid
def test_exclusive_scan_large(n):
    assert torch.cuda.is_available()
    _check_scan(torch.device("cuda:0"), n)


@pytest.mark.parametrize("n,bits", [(2, 8), (1000, 5), (1025, 13), (40001, 32)])
def test_radix_sort_is_stable(backend, n, bits):
    L = _lib.lib()
    rng = np.random.RandomState(n)
    k = rng.randint(0, 2 ** min(bits, 31), n).astype(np.int64)
    if bits == 32:
        k = rng.randint(0, 2 ** 32, n, dtype=np.int64)
    keys = torch.from_numpy(k.astype(np.uint32).view(np.int32)).to(backend)
    vals = torch.arange(n, dtype=torch.int32, device=backend)
    ws = _lib.workspace(L.st_sort_workspace_bytes(n), backend)
    _lib.check(L.st_sort_pairs_u32(_lib.ptr(keys), _lib.ptr(vals), n, bits, _lib.ptr(ws), ws.numel(), _lib.stream(backend)))
    order = np.argsort(k, kind="stable")
    np.testing.assert_array_equal(keys.cpu().numpy().view(np.uint32), k[order].astype(np.uint32))
    np.testing.assert_array_equal(vals.cpu().numpy(), order.astype(np.int32))
