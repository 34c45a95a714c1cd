"""The C-ABI library loads and exports every symbol include/smarttree_hip.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "smarttree_hip.h").read_text()
DECLARED = sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S))))


def test_header_declares_the_boundary():
    for name in ["st_voxelize_blocks", "st_build_subm_rulebook", "st_build_strided_rulebook", "st_sparse_conv_fwd",
                 "st_pointwise_mlp_heads", "st_knn_radius", "st_make_edges", "st_connected_components", "st_sssp",
                 "st_tree_distance", "st_sample_tree", "st_query_workspace", "st_version"]:
        assert name in DECLARED


def test_hip_library_exports_every_declared_symbol():
    from smart_tree_amd import _lib, build_ext

    lib = ctypes.CDLL(str(build_ext.build()))  # hipcc cross-compiles without a GPU
    missing = [n for n in DECLARED if not hasattr(lib, n)]
    assert not missing, missing
    assert set(_lib.SIGNATURES) <= set(DECLARED)
    assert lib.st_version() >= 101
    # the caller-side arrays of the skeleton calls are as long as the library says (advisor, round 5: silent ABI growth)
    from smart_tree_amd.skeleton import tuning
    assert lib.st_abi_entries(0) == 16 == int(re.search(r"#define ST_SKELETON_STATS_ENTRIES (\d+)", HEADER).group(1))
    assert lib.st_abi_entries(1) == int(re.search(r"#define ST_SKELETON_TUNING_ENTRIES (\d+)", HEADER).group(1)) == tuning.ENTRIES
    assert lib.st_abi_entries(2) == 64 and lib.st_abi_entries(99) == -1


def test_product_path_refuses_cpu_tensors():
    import torch

    from smart_tree_amd import _lib

    with pytest.raises(_lib.StError):
        _lib.ptr(torch.zeros(4))
