"""ctypes binding of libsmarttree_hip.so (the C-ABI declared in include/smarttree_hip.h).

The product path has exactly one backend: the HIP library built for gfx950.  If it is missing,
or a tensor that is not on the GPU reaches a kernel wrapper, this module raises -- there is no
CPU fallback.  (tests/hipemu swaps `_LIB` for a CPU *sanitizer build of the same kernel
sources* inside the CPU test-suite only; nothing in this package refers to it.)
"""
from __future__ import annotations

import ctypes
from ctypes import c_double, c_float, c_int, c_int64, c_void_p
from pathlib import Path

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libsmarttree_hip.so"

_LIB = None
_ALLOW_HOST_POINTERS = False  # flipped only by tests/hipemu (CPU sanitizer build of the kernels)

P = c_void_p
I64 = c_int64

# name -> (restype, argtypes); mirrors include/smarttree_hip.h
SIGNATURES = {
    "st_version": (c_int, []),
    "st_abi_entries": (c_int, [c_int]),
    "st_last_error": (ctypes.c_char_p, []),
    "st_scan_workspace_bytes": (I64, [I64]),
    "st_scan_u32": (c_int, [P, P, I64, P, P, I64, P]),
    "st_sort_workspace_bytes": (I64, [I64]),
    "st_sort_pairs_u32": (c_int, [P, P, I64, c_int, P, I64, P]),
    "st_voxelize_workspace_bytes": (I64, [I64, c_int, I64]),
    "st_voxelize_blocks": (c_int, [P, P, I64, c_double, c_double, c_double, c_int, c_int, I64, P, P, P, P, P,
                                   ctypes.POINTER(I64), ctypes.POINTER(I64), P, I64, P]),
    "st_hash_capacity": (I64, [I64]),
    "st_build_coord_hash": (c_int, [P, I64, P, P, I64, P]),
    "st_build_subm_rulebook": (c_int, [P, I64, P, P, I64, P, P]),
    "st_strided_workspace_bytes": (I64, [I64]),
    "st_spatial_order_workspace_bytes": (I64, [I64]),
    "st_spatial_order": (c_int, [P, I64, P, P, I64, P]),
    "st_move_rows": (c_int, [P, c_int, P, I64, P, c_int, P]),
    "st_build_strided_outputs": (c_int, [P, I64, I64, P, P, P, I64, ctypes.POINTER(I64), ctypes.POINTER(ctypes.c_int32),
                                         P, I64, P]),
    "st_build_strided_rulebook": (c_int, [P, I64, P, P, I64, P, I64, P, P, I64, ctypes.POINTER(ctypes.c_int32), P, P, P, P]),
    "st_sparse_conv_fwd": (c_int, [P, c_int, P, c_int, P, c_int, I64, P, c_int, P, P, P, c_int, P, P, P, I64]),
    "st_sparse_conv_mfma_fwd": (c_int, [P, c_int, P, c_int, P, c_int, I64, P, c_int, P, P, P, c_int, P, P, P, I64, c_int]),
    "st_sparse_conv_b3_fwd": (c_int, [P, c_int, P, c_int, P, c_int, I64, P, c_int, P, P, P, c_int, P, P, P, I64, c_int]),
    "st_sparse_conv_f16_fwd": (c_int, [P, c_int, P, c_int, P, c_int, I64, P, c_int, P, P, P, c_int, P, c_int, c_int, P, P, I64]),
    "st_brick_pyramid_workspace_bytes": (I64, [I64, c_int, c_int, c_int, ctypes.POINTER(I64)]),
    "st_brick_pyramid": (c_int, [P, I64, c_int, c_int, c_int, P, c_int, ctypes.POINTER(I64), P, ctypes.POINTER(P), ctypes.POINTER(P),
                                 ctypes.POINTER(P), ctypes.POINTER(P), ctypes.POINTER(P), ctypes.POINTER(I64), P, I64, P]),
    "st_head_param_floats": (c_int, []),
    "st_pointwise_mlp_heads": (c_int, [P, I64, P, P, P, P, P, P, P]),
    "st_knn_workspace_bytes": (I64, [I64]),
    "st_knn_radius": (c_int, [P, I64, P, I64, c_int, c_float, P, c_int, c_float, P, P, P, I64, P]),
    "st_medial_points": (c_int, [P, P, I64, P, P, P]),
    "st_centre_cloud": (c_int, [P, I64, P, P, I64, P]),
    "st_make_edges_workspace_bytes": (I64, [I64]),
    "st_make_edges": (c_int, [P, P, I64, c_int, P, P, ctypes.POINTER(I64), P, I64, P]),
    "st_connected_components_workspace_bytes": (c_int64, [c_int64]),
    "st_connected_components": (c_int, [P, I64, I64, P, P, I64, P]),
    "st_component_layout_workspace_bytes": (I64, [I64]),
    "st_component_layout": (c_int, [P, I64, c_int, P, P, P, P, ctypes.POINTER(I64), ctypes.POINTER(I64), P, I64, P]),
    "st_component_csr_workspace_bytes": (I64, [I64]),
    "st_component_csr": (c_int, [P, P, I64, P, I64, P, P, P, P, I64, P]),
    "st_assemble_workspace_bytes": (c_int64, [c_int64]),
    "st_assemble_branches": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, P, P, c_int64, P]),
    "st_points_to_nearest_tube": (c_int, [P, c_int64, P, P, P, P, c_int64, P, P, P, P]),
    "st_post_process": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, c_float, c_float, c_int, c_int, c_int, P]),
    "st_skeleton_workspace_bytes": (I64, [I64, I64]),
    # batched forms (B clouds per launch set)
    "st_centre_cloud_seg": (c_int, [P, I64, P, c_int, P, P, I64, P]),
    "st_voxelize_workspace_bytes_seg": (I64, [I64, c_int, I64, c_int]),
    "st_voxelize_blocks_seg": (c_int, [P, P, I64, P, c_int, c_double, c_double, c_double, c_int, c_int, I64, P, P, P, P, P, P, P, P,
                                       ctypes.POINTER(I64), ctypes.POINTER(I64), P, I64, P]),
    "st_voxelize_cloud_workspace_bytes": (I64, [I64, I64, c_int]),
    "st_voxelize_cloud_seg": (c_int, [P, P, I64, P, c_int, c_double, I64, P, P, P, P, P, ctypes.POINTER(I64), P, I64, P]),
    "st_loss_workspace_bytes": (I64, []),
    "st_loss_forward": (c_int, [P, P, P, c_int, P, c_int, P, I64, c_int, c_int, ctypes.POINTER(ctypes.c_double), P, I64, P]),
    "st_build_strided_outputs_seg": (c_int, [P, I64, I64, P, P, P, I64, ctypes.POINTER(I64), ctypes.POINTER(ctypes.c_int32),
                                             P, c_int, P, P, I64, P]),
    "st_build_strided_rulebook_seg": (c_int, [P, I64, P, P, I64, P, I64, P, P, I64, ctypes.POINTER(ctypes.c_int32), P, P, P, P, P, P]),
    "st_knn_workspace_bytes_seg": (I64, [I64, c_int]),
    "st_knn_radius_seg": (c_int, [P, I64, P, I64, c_int, c_float, P, c_int, c_float, P, P, P, P, c_int, P, I64, P, c_float]),
    "st_radius_count_seg": (c_int, [P, I64, P, I64, c_int, c_float, P, c_int, c_float, P, P, P, c_int, P, I64, P, c_float, P]),
    "st_connected_components_knn": (c_int, [P, I64, c_int, P, P, P, I64, P]),
    "st_component_csr_knn": (c_int, [P, P, I64, c_int, P, P, I64, P, P, P, P, I64, P]),
    "st_component_csr_knn_workspace_bytes": (I64, [I64, I64, c_int]),
    "st_make_edges_seg": (c_int, [P, P, I64, c_int, P, P, ctypes.POINTER(I64), P, c_int, P, I64, P]),
    "st_component_layout_seg": (c_int, [P, I64, c_int, P, c_int, P, P, P, P, P, P, P, ctypes.POINTER(I64), ctypes.POINTER(I64), P, I64, P,
                                        ctypes.POINTER(I64)]),
    "st_skeleton_workspace_bytes_seg": (I64, [I64, I64, c_int]),
    "st_skeleton_components_seg": (c_int, [c_int, P, P, P, c_int, I64, P, P, P, P, P, P, c_float, c_int, c_int, P, P, P, P, P, P, P, P,
                                           P, P, ctypes.POINTER(I64), P, I64, P, ctypes.POINTER(I64)]),
    "st_post_process_seg": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, c_float, c_float, c_int, c_int, c_int, P, c_int, P]),
    "st_skeleton_components": (c_int, [c_int, P, P, I64, P, P, P, P, P, P, c_float, c_int, c_int, P, P, P, P, P, P, P, P,
                                       P, P, ctypes.POINTER(I64), P, I64, P]),
}


# Entry points that only ENQUEUE work (or compute a size) and return in microseconds.  ctypes drops the GIL around
# every CDLL call; with several clouds in flight (one host thread each, bench.py --streams) re-acquiring it after a
# 5 us launch means queueing behind the other threads -- a convoy that costs more than the launch.  These are bound
# through PyDLL (GIL kept); everything that waits for the GPU inside the call (a count read back) stays on CDLL.
ENQUEUE_ONLY = frozenset({
    "st_version", "st_abi_entries", "st_last_error", "st_scan_workspace_bytes", "st_sort_workspace_bytes", "st_voxelize_workspace_bytes",
    "st_hash_capacity", "st_strided_workspace_bytes", "st_head_param_floats", "st_knn_workspace_bytes",
    "st_make_edges_workspace_bytes", "st_connected_components_workspace_bytes", "st_component_layout_workspace_bytes",
    "st_component_csr_workspace_bytes", "st_assemble_workspace_bytes", "st_skeleton_workspace_bytes",
    "st_build_coord_hash", "st_build_subm_rulebook", "st_build_strided_rulebook", "st_sparse_conv_fwd",
    "st_sparse_conv_mfma_fwd", "st_sparse_conv_b3_fwd", "st_sparse_conv_f16_fwd", "st_pointwise_mlp_heads", "st_medial_points", "st_centre_cloud",
    "st_connected_components", "st_component_csr", "st_post_process", "st_knn_radius", "st_brick_pyramid_workspace_bytes",
    "st_centre_cloud_seg", "st_voxelize_workspace_bytes_seg", "st_build_strided_rulebook_seg", "st_knn_workspace_bytes_seg",
    "st_knn_radius_seg", "st_skeleton_workspace_bytes_seg", "st_post_process_seg", "st_radius_count_seg",
    "st_voxelize_cloud_workspace_bytes", "st_loss_workspace_bytes", "st_spatial_order_workspace_bytes", "st_spatial_order", "st_connected_components_knn", "st_component_csr_knn", "st_component_csr_knn_workspace_bytes", "st_move_rows",
})


# Entry points that wait only when asked for a count on the host: `<name>_nowait` is the GIL-keeping binding for calls
# that pass NULL for it (st_make_edges: n_edges_host, st_assemble_branches: counts_host).
NOWAIT_VARIANTS = ("st_make_edges", "st_assemble_branches", "st_make_edges_seg")


class _Bound:
    """The declared entry points as attributes (one ctypes function object each); any other exported symbol
    resolves through the GIL-dropping handle."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):  # only reached for names that were not bound by declare()
        return getattr(self._cdll, name)


def declare(cdll, pydll=None):
    """Set the C signatures on `cdll`; returns the object whose attributes are the callable entry points.  With
    `pydll` (a second handle of the SAME library that keeps the GIL) the ENQUEUE_ONLY functions are taken from it."""
    out = _Bound(cdll) if pydll is not None else cdll
    for name, (res, args) in SIGNATURES.items():
        for handle in ((cdll, pydll) if pydll is not None else (cdll,)):
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if pydll is not None:
            setattr(out, name, getattr(pydll if name in ENQUEUE_ONLY else cdll, name))
    for name in NOWAIT_VARIANTS:
        setattr(out, name + "_nowait", getattr(pydll if pydll is not None else cdll, name))
    return out


def lib():
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "smart_tree_amd has no CPU fallback.")
        import os
        if os.environ.get("ST_CDLL_ONLY") == "1":  # developer knob for A/B timing of the GIL policy
            _LIB = declare(ctypes.CDLL(str(LIB_PATH)))
        else:
            _LIB = declare(ctypes.CDLL(str(LIB_PATH)), ctypes.PyDLL(str(LIB_PATH)))
    return _LIB


class StError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise StError(f"smarttree_hip error {rc}: {lib().st_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda and not _ALLOW_HOST_POINTERS:
        raise StError("smart_tree_amd kernels need tensors on the GPU (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise StError("smart_tree_amd kernels need contiguous tensors")
    return t.data_ptr()


_HAS_GPU = None
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream(device=None):
    """Raw handle of torch's current HIP stream on `device` (the calling thread's: streams are per thread)."""
    global _HAS_GPU
    if _HAS_GPU is None:
        _HAS_GPU = torch.cuda.is_available()
    if _HAS_GPU and not _ALLOW_HOST_POINTERS:
        if _RAW_STREAM is not None:  # one C call instead of building a torch.cuda.Stream object per kernel launch
            idx = device.index if isinstance(device, torch.device) and device.index is not None else torch.cuda.current_device()
            return _RAW_STREAM(idx)
        return torch.cuda.current_stream(device).cuda_stream
    return None


def workspace(nbytes: int, device) -> torch.Tensor:
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    if _ALLOW_HOST_POINTERS:
        ws.fill_(0xCD)  # sanitizer build only: poison scratch so reads of never-written words cannot pass by luck
    return ws
