"""Thin torch-tensor wrappers over the C-ABI rulebook / conv entry points (csrc/rulebook.hip,
csrc/sparse_conv.hip).  Every function allocates its outputs with torch (plumbing only) and
enqueues on the current HIP stream."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .. import _lib, profiling


@dataclass
class CoordHash:
    keys: torch.Tensor  # [cap] int64 (bit pattern of the u64 keys)
    vals: torch.Tensor  # [cap] int32
    cap: int


def build_coord_hash(coords: torch.Tensor) -> CoordHash:
    L = _lib.lib()
    n = coords.shape[0]
    cap = L.st_hash_capacity(n)
    keys = torch.empty(cap, dtype=torch.int64, device=coords.device)
    vals = torch.empty(cap, dtype=torch.int32, device=coords.device)
    _lib.check(L.st_build_coord_hash(_lib.ptr(coords), n, _lib.ptr(keys), _lib.ptr(vals), cap, _lib.stream(coords.device)))
    return CoordHash(keys, vals, cap)


def build_subm_rulebook(coords: torch.Tensor, h: CoordHash) -> torch.Tensor:
    """nbr [27, N] int32: input row at o + (k-1) or -1."""
    L = _lib.lib()
    n = coords.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=coords.device)
    _lib.check(L.st_build_subm_rulebook(_lib.ptr(coords), n, _lib.ptr(h.keys), _lib.ptr(h.vals), h.cap, _lib.ptr(nbr),
                                        _lib.stream(coords.device)))
    return nbr


@dataclass
class StridedRulebook:
    out_coords: torch.Tensor  # [M,4] int32 coarse active set, canonical order
    out_hash: CoordHash
    nbr_down: torch.Tensor  # [27, M] fine rows
    nbr_up: torch.Tensor  # [27, N] coarse rows
    up_order: torch.Tensor = None  # [N] fine rows grouped by coordinate parity: the inverse conv's launch order


def build_strided_rulebook(coords: torch.Tensor, h: CoordHash, blk_seg: Optional[torch.Tensor] = None,
                           n_seg: int = 1) -> StridedRulebook:
    """blk_seg / n_seg (batched clouds): the output set of every cloud is clipped to that cloud's own spatial extent."""
    L = _lib.lib()
    dev = coords.device
    n = coords.shape[0]
    if blk_seg is None or blk_seg.numel() == 0:  # (a batch without a single block: nothing to clip)
        n_seg = 1
    ext_dev = torch.empty(3 * n_seg, dtype=torch.int32, device=dev) if n_seg > 1 else None
    ws = _lib.workspace(L.st_strided_workspace_bytes(n), dev)
    n_out = ctypes.c_int64(0)
    extent = (ctypes.c_int32 * 3)()
    rc = -1
    for max_out in (n + 1024, 8 * n + 1024):  # k3 s2 p1: an input reaches <= 8 outputs; dense data halves the set
        out_coords = torch.empty((max_out, 4), dtype=torch.int32, device=dev)
        ccap = L.st_hash_capacity(max_out)
        ckeys = torch.empty(ccap, dtype=torch.int64, device=dev)
        cvals = torch.empty(ccap, dtype=torch.int32, device=dev)
        rc = L.st_build_strided_outputs_seg(_lib.ptr(coords), n, max_out, _lib.ptr(out_coords), _lib.ptr(ckeys),
                                            _lib.ptr(cvals), ccap, ctypes.byref(n_out), extent, _lib.ptr(blk_seg) if n_seg > 1 else None,
                                            n_seg, _lib.ptr(ext_dev), _lib.ptr(ws), ws.numel(), _lib.stream(dev))
        if rc == 0 or b"max_out" not in L.st_last_error():
            break
    _lib.check(rc)
    m = n_out.value
    out_coords = out_coords[:m].contiguous()
    nbr_down = torch.empty((27, m), dtype=torch.int32, device=dev)
    nbr_up = torch.empty((27, n), dtype=torch.int32, device=dev)
    up_order = torch.empty(n + 16, dtype=torch.int32, device=dev)  # the 16-word tail is the kernel's scratch
    _lib.check(L.st_build_strided_rulebook_seg(_lib.ptr(coords), n, _lib.ptr(h.keys), _lib.ptr(h.vals), h.cap,
                                               _lib.ptr(out_coords), m, _lib.ptr(ckeys), _lib.ptr(cvals), ccap, extent,
                                               _lib.ptr(nbr_down), _lib.ptr(nbr_up), _lib.ptr(up_order),
                                               _lib.ptr(blk_seg) if n_seg > 1 else None, _lib.ptr(ext_dev), _lib.stream(dev)))
    return StridedRulebook(out_coords, CoordHash(ckeys, cvals, ccap), nbr_down, nbr_up, up_order[:n])


@dataclass
class RulebookPyramid:
    """Per-level active sets and neighbour tables of one batch; built once, shared by all convs."""
    coords: List[torch.Tensor] = field(default_factory=list)
    subm: List[torch.Tensor] = field(default_factory=list)
    down: List[torch.Tensor] = field(default_factory=list)  # down[l]: level l -> l+1
    up: List[torch.Tensor] = field(default_factory=list)  # up[l]:   level l+1 -> l
    up_order: List[torch.Tensor] = field(default_factory=list)  # up_order[l]: launch order of level l's rows for up[l]


def build_pyramid(coords: torch.Tensor, depth: int, blk_seg: Optional[torch.Tensor] = None, n_seg: int = 1) -> RulebookPyramid:
    pyr = RulebookPyramid()
    h = build_coord_hash(coords)
    for level in range(depth + 1):
        pyr.coords.append(coords)
        pyr.subm.append(build_subm_rulebook(coords, h))
        if level == depth:
            break
        s = build_strided_rulebook(coords, h, blk_seg, n_seg)
        pyr.down.append(s.nbr_down)
        pyr.up.append(s.nbr_up)
        pyr.up_order.append(s.up_order)
        coords, h = s.out_coords, s.out_hash
    return pyr


def brick_pyramid(coords: torch.Tensor, depth: int, n_blocks: int, coord_bound: int, blk_seg: Optional[torch.Tensor] = None,
                  n_seg: int = 1):
    """The whole pyramid from occupancy bricks + popcount ranks (csrc/brick.hip): no hash probes, no sorts, ONE read-back.
    Returns (RulebookPyramid, order0) or None when the structure cannot be sized for (n_blocks, coord_bound) -- the caller
    then takes `build_pyramid`.  Level 0's rows are the input voxels in brick order (row p = input voxel order0[p]); the
    tables are views with row stride = the level's capacity (`sparse_conv` passes the stride on)."""
    L = _lib.lib()
    dev = coords.device
    n0 = coords.shape[0]
    if n0 == 0 or n_blocks < 1 or coord_bound < 1:
        return None
    if blk_seg is None or blk_seg.numel() == 0:
        n_seg = 1
    i32 = dict(dtype=torch.int32, device=dev)
    for attempt in range(2):
        # k3 s2 p1: a surface-like set shrinks to ~half per level.  Isolated voxels reach up to 8 outputs each and thin lines
        # ~1.1 per fine voxel: the retry sizes every level for 8x the level below, clamped by what the declared geometry can
        # hold (blocks x cells of the level's grid); tables of that size that do not fit a budget (27 x 4 bytes x three tables
        # per row) are not attempted -- the caller then takes the hash-table builders (build_pyramid), which size themselves
        # level by level from exact counts.
        caps = [n0]
        for l in range(depth):
            side = -(-int(coord_bound) // (2 << l)) + 1  # ceil(bound / 2^(l+1)) (+1: padding 1 reaches one cell further)
            geom = int(n_blocks) * side ** 3
            caps.append((3 * caps[-1]) // 4 + 4096 if attempt == 0 else min(8 * caps[-1] + 4096, geom))
        if attempt == 1 and sum(caps) * 27 * 4 * 3 > (16 << 30):
            return None
        c_caps = (ctypes.c_int64 * (depth + 1))(*caps)
        nbytes = L.st_brick_pyramid_workspace_bytes(n0, int(n_blocks), int(coord_bound), depth, c_caps)
        if nbytes <= 0 or nbytes > (24 << 30):
            return None
        try:
            ws = _lib.workspace(nbytes, dev)
            order0 = torch.empty(n0, **i32)
            coords_out = [torch.empty((caps[l], 4), **i32) for l in range(depth + 1)]
            subm = [torch.empty((27, caps[l]), **i32) for l in range(depth + 1)]
            down = [torch.empty((27, caps[l + 1]), **i32) for l in range(depth)]
            up = [torch.empty((27, caps[l]), **i32) for l in range(depth)]
            up_order = [torch.empty(caps[l] + 16, **i32) for l in range(depth)]
        except torch.OutOfMemoryError:
            return None
        arr = lambda ts: (ctypes.c_void_p * max(len(ts), 1))(*[t.data_ptr() for t in ts])
        counts = (ctypes.c_int64 * (depth + 1))()
        rc = L.st_brick_pyramid(_lib.ptr(coords), n0, int(n_blocks), int(coord_bound), depth, _lib.ptr(blk_seg) if n_seg > 1 else None,
                                n_seg, c_caps, _lib.ptr(order0), arr(coords_out), arr(subm), arr(down), arr(up), arr(up_order), counts,
                                _lib.ptr(ws), ws.numel(), _lib.stream(dev))
        if rc == 0:
            break
        if b"outside the declared bounds" in L.st_last_error():
            return None  # the hint was wrong for this input: hash-table builders
        if b"capacity" not in L.st_last_error():
            _lib.check(rc)
        if attempt == 1:
            return None  # still over capacity: hash-table builders
        del ws, order0, coords_out, subm, down, up, up_order
    n = [int(counts[l]) for l in range(depth + 1)]
    pyr = RulebookPyramid()
    for l in range(depth + 1):
        pyr.coords.append(coords_out[l][: n[l]])
        pyr.subm.append(subm[l][:, : n[l]])
        if l < depth:
            pyr.down.append(down[l][:, : n[l + 1]])
            pyr.up.append(up[l][:, : n[l]])
            pyr.up_order.append(up_order[l][: n[l]])
    return pyr, order0


def _nbr_args(nbr: Optional[torch.Tensor]):
    """(device pointer, row stride) of a neighbour table [K, n] -- contiguous, or a column slice of a capacity-strided one."""
    if nbr is None:
        return None, 0
    if nbr.ndim != 2 or (nbr.shape[1] > 1 and nbr.stride(1) != 1):
        raise _lib.StError("neighbour table must be [K, n] with unit column stride")
    if not nbr.is_cuda and not _lib._ALLOW_HOST_POINTERS:
        raise _lib.StError("smart_tree_amd kernels need tensors on the GPU (got a CPU tensor); there is no CPU fallback")
    return nbr.data_ptr(), int(nbr.stride(0)) if nbr.shape[0] > 1 else int(max(nbr.shape[1], 1))


B3_VARIANT = 0  # bench aid: row tiles per wavefront of the split-bf16 kernel (0 = by size)
MFMA_VARIANT = 0  # bench aid (tools/bench_conv.py): tile shape of the f32 matrix-core kernel, 0 = the library picks by size


def spatial_order(coords: torch.Tensor) -> torch.Tensor:
    """[N] int32 permutation: voxels sorted by (batch index, Morton code of z, y, x) (csrc/rulebook.hip st_spatial_order)."""
    L = _lib.lib()
    n = coords.shape[0]
    order = torch.empty(n, dtype=torch.int32, device=coords.device)
    if n:
        ws = _lib.workspace(L.st_spatial_order_workspace_bytes(n), coords.device)
        _lib.check(L.st_spatial_order(_lib.ptr(coords), n, _lib.ptr(order), _lib.ptr(ws), ws.numel(), _lib.stream(coords.device)))
    return order


def move_rows(x: torch.Tensor, order: torch.Tensor, scatter: bool = False) -> torch.Tensor:
    """x[order] (gather) or the inverse (out[order] = x, scatter) for a [N, C] tensor of 4-byte elements, order int32 [N]."""
    L = _lib.lib()
    x = x.contiguous()
    assert x.element_size() == 4 and x.ndim == 2 and order.dtype == torch.int32
    out = torch.empty_like(x)
    _lib.check(L.st_move_rows(_lib.ptr(x), x.shape[1], _lib.ptr(order), x.shape[0], _lib.ptr(out), int(scatter), _lib.stream(x.device)))
    return out


def mfma_weight(w: torch.Tensor) -> torch.Tensor:
    """[K, Cin, Cout] -> the MFMA operand order wp[K][Cin/16][4][Cout][4] = W[k][16c + 4kg + s][co]."""
    K, cin, cout = w.shape
    return w.reshape(K, cin // 16, 4, 4, cout).permute(0, 1, 2, 4, 3).contiguous()


def mfma_weight32(w: torch.Tensor) -> torch.Tensor:
    """[K, Cin, Cout] (Cin % 32 == 0) -> the 32-deep operand order wp[K][Cin/32][4][Cout][8] = W[k][32c + 8g + e][co]: what
    st_sparse_conv_f16_fwd expects (as half) for Cin % 32 == 0; 16-channel inputs keep mfma_weight's order."""
    K, cin, cout = w.shape
    return w.reshape(K, cin // 32, 4, 8, cout).permute(0, 1, 2, 4, 3).contiguous()


def mfma_weight16_half(w: torch.Tensor) -> torch.Tensor:
    """The half-precision matrix-core weights of a conv in the order its kernel reads them: 32-channel chunks (mfma_weight32);
    16 input channels with 16 / 32 outputs: pairs of kernel offsets stacked into 32-channel chunks (zeros behind an odd last
    offset); anything else: mfma_weight's 16-channel order."""
    K, cin, cout = w.shape
    if cin % 32 == 0:
        return mfma_weight32(w).half()
    if cin == 16 and cout in (16, 32):
        if K % 2:
            w = torch.cat([w, torch.zeros((1, cin, cout), dtype=w.dtype, device=w.device)], 0)
        return mfma_weight32(w.reshape((K + 1) // 2, 32, cout)).half()
    return mfma_weight(w).half()


_MFMA_SHAPES = ((16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64))  # the instantiated (Cin, Cout) of the matrix-core kernels


def mfma_eligible(cin: int, cout: int, c0: int) -> bool:
    """The matrix-core kernels exist for the shipped architecture's widths; any other (Cin, Cout) takes st_sparse_conv_fwd, whose
    generic kernel handles every width."""
    return (cin, cout) in _MFMA_SHAPES and c0 % 16 == 0


def b3_eligible(cin: int, cout: int, c0: int) -> bool:
    """The split-bf16 matrix-core kernel (st_sparse_conv_b3_fwd): 32-channel chunks (16 input channels: two kernel offsets per
    chunk, no concat), 16-channel column tiles."""
    if cin == 16:
        return c0 == 16 and cout in (16, 32)
    return cin % 32 == 0 and cout % 16 == 0 and c0 % 8 == 0 and (cin, cout) in ((32, 16), (32, 32), (32, 64), (64, 32), (64, 64))


def b3_weight(w: torch.Tensor) -> torch.Tensor:
    """[K, Cin, Cout] float32 -> three bf16 planes in operand order (int16 bit patterns):
    Cin % 32 == 0: wq[K][Cin/32][3][4][Cout][8] = piece p of W[k][32c + 8g + e][co];
    Cin == 16:     wq[ceil(K/2)][3][4][Cout][8] = piece p of W[2j + (g >> 1)][8 (g & 1) + e][co] (two offsets per 32-deep chunk, zeros
                   behind an odd last offset).
    hi = the upper 16 bits of the float, mid = those of w - hi, lo = those of w - hi - mid -- truncating splits, so
    hi + mid + lo == w exactly (csrc/sparse_conv.hip "split-bf16 rule-GEMM")."""
    K, cin, cout = w.shape
    w = w.detach().to(torch.float32).contiguous()
    if cin == 16:  # pairs of offsets stacked along the channel axis
        if K % 2:
            w = torch.cat([w, torch.zeros((1, cin, cout), dtype=w.dtype, device=w.device)], 0)
        w = w.reshape((K + 1) // 2, 32, cout)
        K, cin = w.shape[0], 32
    mask = torch.tensor(-65536, dtype=torch.int32, device=w.device)  # 0xffff0000
    planes, r = [], w
    for _ in range(3):
        top = (r.view(torch.int32) & mask).view(torch.float32)
        planes.append((r.view(torch.int32) >> 16).to(torch.int16))  # the upper 16 bits (arithmetic shift keeps the bit pattern)
        r = r - top  # exact
    q = torch.stack(planes, 0)  # [3, K, Cin, Cout]
    q = q.reshape(3, K, cin // 32, 4, 8, cout).permute(1, 2, 0, 3, 5, 4).contiguous()  # [K, c, plane, g, co, e]
    return q


def sparse_conv(x0: torch.Tensor, w: torch.Tensor, nbr: Optional[torch.Tensor], n_out: int,
                x1: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
                shift: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                relu: bool = False, wp: Optional[torch.Tensor] = None, out_half: bool = False,
                wp16: Optional[torch.Tensor] = None, row_order: Optional[torch.Tensor] = None,
                wq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(bn(sum_k W_k . cat(x0, x1)[nbr[k]]) + residual); w is [K, Cin, Cout].
    wp (optional): the same weights in MFMA order -> the matrix-core kernel is used.
    wq (optional, b3_weight(w)): the weights as three bf16 planes -> the split-bf16 matrix-core kernel (float32 accuracy on the
    bf16 pipe) where the shape is eligible (b3_eligible); takes precedence over wp.
    row_order (optional, [n_out] int32): launch order of the output rows (never changes the result).
    Half-precision storage mode: a float16 x0 and / or out_half select st_sparse_conv_f16_fwd (wp16 = mfma_weight16_half(w): the layout depends on Cin / Cout, see include/smarttree_hip.h)."""
    L = _lib.lib()
    K, cin, cout = w.shape
    c0 = x0.shape[1]
    nbr_ptr, nbr_stride = _nbr_args(nbr)
    if n_out == 0:  # an empty active set (a cloud without a block of > 20 points): nothing to launch
        return torch.empty((0, cout), dtype=torch.float16 if out_half else torch.float32, device=x0.device)
    in_half = x0.dtype == torch.float16
    if in_half or out_half:
        both = in_half and out_half
        if both and wp16 is None:
            raise ValueError("half -> half convolution needs the half-precision MFMA weights (wp16)")
        y = torch.empty((n_out, cout), dtype=torch.float16 if out_half else torch.float32, device=x0.device)
        esz_in, esz_out = (2 if in_half else 4), (2 if out_half else 4)
        nbytes = (lambda: (_pair_count(nbr) if nbr is not None else n_out) * (cin * esz_in + (4 if nbr is not None else 0))
                  + n_out * cout * esz_out) if profiling.enabled() else 0
        nflops = (lambda: 2.0 * (_pair_count(nbr) if nbr is not None else n_out) * cin * cout) if profiling.enabled() else 0
        name = f"k_sparse_conv_mfma_f16<{cin},{cout}>" if both else f"k_sparse_conv<{cin},{cout}> {'h->f' if in_half else 'f->h'}"
        with profiling.kernel(name + ("" if nbr is not None else " k1"), nbytes, nflops):
            _lib.check(L.st_sparse_conv_f16_fwd(_lib.ptr(x0), c0, _lib.ptr(x1), cin, nbr_ptr, K, n_out,
                                                _lib.ptr(wp16 if both else w), cout, _lib.ptr(scale), _lib.ptr(shift),
                                                _lib.ptr(residual), int(relu), _lib.ptr(y), int(in_half), int(out_half),
                                                _lib.ptr(row_order), _lib.stream(x0.device), nbr_stride))
        return y
    # 16 -> 16 submanifold convs run on the vector kernel (lane = voxel, weights from scalar registers): with Morton-ordered
    # rows it beats the matrix-core kernel wherever the level fills the chip (67 % against 53 % of the HBM peak at 1.5M rows,
    # profiles/r02_conv_layers_batch16.txt).  At EVERY size, not only the large ones: the two kernels round differently in
    # the last bit (the matrix core does not evaluate a k-ordered fmaf chain exactly), and a cloud must get the same values
    # alone and inside a batch (tests/test_batch.py).
    if wq is not None and b3_eligible(cin, cout, c0):
        y = torch.empty((n_out, cout), dtype=torch.float32, device=x0.device)
        nbytes = (lambda: (_pair_count(nbr) if nbr is not None else n_out) * (cin * 4 + (4 if nbr is not None else 0))
                  + n_out * cout * 4) if profiling.enabled() else 0
        nflops = (lambda: 2.0 * (_pair_count(nbr) if nbr is not None else n_out) * cin * cout) if profiling.enabled() else 0
        with profiling.kernel(f"k_sparse_conv_mfma_b3<{cin},{cout}>" + ("" if nbr is not None else " k1"), nbytes, nflops):
            _lib.check(L.st_sparse_conv_b3_fwd(_lib.ptr(x0), c0, _lib.ptr(x1), cin, nbr_ptr, K, n_out, _lib.ptr(wq), cout,
                                               _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), int(relu), _lib.ptr(y),
                                               _lib.ptr(row_order), _lib.stream(x0.device), nbr_stride, int(B3_VARIANT)))
        return y
    big_16 = cin == 16 and cout == 16 and nbr is not None and row_order is None and x1 is None
    if wp is not None and mfma_eligible(cin, cout, c0) and not big_16:
        y = torch.empty((n_out, cout), dtype=torch.float32, device=x0.device)
        nbytes = (lambda: (_pair_count(nbr) if nbr is not None else n_out) * (cin * 4 + (4 if nbr is not None else 0))
                  + n_out * cout * 4) if profiling.enabled() else 0
        nflops = (lambda: 2.0 * (_pair_count(nbr) if nbr is not None else n_out) * cin * cout) if profiling.enabled() else 0
        with profiling.kernel(f"k_sparse_conv_mfma<{cin},{cout}>" + ("" if nbr is not None else " k1"), nbytes, nflops):
            _lib.check(L.st_sparse_conv_mfma_fwd(_lib.ptr(x0), c0, _lib.ptr(x1), cin, nbr_ptr, K, n_out, _lib.ptr(wp), cout,
                                                 _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), int(relu), _lib.ptr(y),
                                                 _lib.ptr(row_order), _lib.stream(x0.device), nbr_stride, int(MFMA_VARIANT)))
        return y
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x0.device)
    # algorithmic bytes: P*(Cin*4 + 4) + N*Cout*4 (pointwise: N*(Cin+Cout)*4); the pair count
    # is evaluated lazily, after the timed region
    nbytes = (lambda: (_pair_count(nbr) if nbr is not None else n_out) * (cin * 4 + (4 if nbr is not None else 0))
              + n_out * cout * 4) if profiling.enabled() else 0
    nflops = (lambda: 2.0 * (_pair_count(nbr) if nbr is not None else n_out) * cin * cout) if profiling.enabled() else 0
    with profiling.kernel(f"k_sparse_conv<{cin},{cout}>" + ("" if nbr is not None else " k1"), nbytes, nflops):
      _lib.check(L.st_sparse_conv_fwd(_lib.ptr(x0), c0, _lib.ptr(x1), cin, nbr_ptr, K, n_out, _lib.ptr(w), cout,
                                    _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), int(relu), _lib.ptr(y),
                                    _lib.ptr(row_order), _lib.stream(x0.device), nbr_stride))
    return y


_pairs_cache = {}


def _pair_count(nbr: torch.Tensor) -> int:
    """Active (input, output) pairs of a neighbour table (profiling only; cached per table)."""
    key = (id(nbr), profiling.generation())  # the thunk keeps the tensor alive, so the id is stable within one collection
    if key not in _pairs_cache:
        if len(_pairs_cache) > 4096:
            _pairs_cache.clear()
        _pairs_cache[key] = int((nbr >= 0).sum().item())
    return _pairs_cache[key]


def mlp_heads(x: torch.Tensor, params: torch.Tensor, with_tail: bool = False):
    L = _lib.lib()
    n, dev = x.shape[0], x.device
    radius = torch.empty((n, 1), dtype=torch.float32, device=dev)
    direction = torch.empty((n, 3), dtype=torch.float32, device=dev)
    class_l = torch.empty((n, 2), dtype=torch.float32, device=dev)
    mv = torch.empty((n, 3), dtype=torch.float32, device=dev) if with_tail else None
    cls = torch.empty((n, 1), dtype=torch.int64, device=dev) if with_tail else None
    if n == 0:
        return radius, direction, class_l, mv, cls
    _lib.check(L.st_pointwise_mlp_heads(_lib.ptr(x), n, _lib.ptr(params), _lib.ptr(radius), _lib.ptr(direction),
                                        _lib.ptr(class_l), _lib.ptr(mv), _lib.ptr(cls), _lib.stream(dev)))
    return radius, direction, class_l, mv, cls
