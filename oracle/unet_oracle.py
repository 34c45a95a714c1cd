"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference network as the shipped checkpoints define it
(the graph is rebuilt from state_dict key names, never unpickled):

  Smart_Tree.forward        smart_tree/model/model.py:77-87
  SubMConvBlock             smart_tree/model/model_blocks.py:8-38      (input conv k1 3->8 + BN + ReLU)
  ResBlock                  smart_tree/model/model_blocks.py:107-156
  EncoderBlock/DecoderBlock smart_tree/model/model_blocks.py:41-104
  UBlock                    smart_tree/model/model_blocks.py:159-243   (skip copy, cat(skip, decoded))
  SparseFC heads            smart_tree/model/model_blocks.py:246-285   (what the checkpoints hold)
  ModelInference tail       smart_tree/model/model_inference.py:87-88  (exp(radius)*direction, argmax)

Third-party arithmetic restated here (spconv-cu117, unpinned, source not under /root/reference
-- parity UNPINNED; the restatement is cross-checked against dense torch conv3d /
conv_transpose3d in tests/test_oracle_unet.py):
  * SubMConv3d k3: out[o] = sum_k W[:,kz,ky,kx,:] . in[i],  i = o + (k - 1), k = (kz*3+ky)*3+kx,
    output set = input set; weights are [Cout,kz,ky,kx,Cin] (checkpoint shapes).
  * SparseConv3d k3 s2 p1: output set = every o with i = 2o - 1 + k active, 0 <= o < out_shape,
    out_shape = (S - 1)//2 + 1.  Output ORDER is hash-dependent in spconv; the canonical order
    here is first appearance when inputs are scanned in index order and k ascending.
  * SparseInverseConv3d k3: re-uses the strided pairs (i,o,k) with roles swapped, same k:
    out[i] += W[:,kz,ky,kx,:] . in[o].
  * spatial extent: the reference passes `spatial_shape = max(coords)` (sparse.py:15-18, not +1),
    which makes spconv treat the max-face voxels as out of range (aliasing hash keys).  That is
    undefined behaviour of the third-party library and depends on the random batch composition;
    the oracle (and the HIP path) use the true extent max+1 over the whole batch -- documented
    deviation, DESIGN.md "Quirks".
  * BatchNorm1d eval, eps = 1e-4 (checkpoint attribute; model.py:23 would default to 1e-5).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


# ----------------------------------------------------------------------------- rulebooks ---
def _extent(coords: np.ndarray):
    return tuple(int(v) + 1 for v in coords[:, 1:].max(0)) if len(coords) else (1, 1, 1)


def _keys(coords: np.ndarray, shape) -> np.ndarray:
    c = coords.astype(np.int64)
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


class _Lookup:
    """coordinate -> row index (or -1) over a fixed active set."""

    def __init__(self, coords: np.ndarray, shape):
        self.shape = shape
        k = _keys(coords, shape)
        self.order = np.argsort(k, kind="stable")
        self.sorted = k[self.order]

    def __call__(self, coords: np.ndarray) -> np.ndarray:
        shape = np.asarray(self.shape)
        inside = np.all((coords[:, 1:] >= 0) & (coords[:, 1:] < shape), axis=1)
        k = _keys(np.where(inside[:, None], coords, 0), self.shape)
        pos = np.searchsorted(self.sorted, k)
        pos = np.minimum(pos, len(self.sorted) - 1) if len(self.sorted) else pos
        hit = inside & (len(self.sorted) > 0)
        if len(self.sorted):
            hit &= self.sorted[pos] == k
        out = np.full(len(coords), -1, np.int64)
        out[hit] = self.order[pos[hit]]
        return out


_OFFS = np.array([(kz, ky, kx) for kz in range(3) for ky in range(3) for kx in range(3)], dtype=np.int64)


def subm_rulebook(coords: np.ndarray) -> np.ndarray:
    """nbr[k, o] = input row at o + (k-1) or -1.  coords [N,4] (b,z,y,x)."""
    look = _Lookup(coords, _extent(coords))
    nbr = np.empty((27, len(coords)), np.int64)
    for k, off in enumerate(_OFFS):
        q = coords.astype(np.int64).copy()
        q[:, 1:] += off - 1
        nbr[k] = look(q)
    return nbr


def strided_out_coords(coords: np.ndarray) -> np.ndarray:
    """Active set of SparseConv3d(k3,s2,p1) in canonical first-appearance order."""
    n = len(coords)
    shape = np.asarray(_extent(coords))
    out_shape = (shape - 1) // 2 + 1
    c = coords.astype(np.int64)
    cand = np.repeat(c, 27, axis=0)  # (i major, k minor)
    num = cand[:, 1:] + 1 - np.tile(_OFFS, (n, 1))  # 2*o
    o = num // 2
    ok = np.all((num % 2 == 0) & (o >= 0) & (o < out_shape), axis=1)
    cand[:, 1:] = o
    cand = cand[ok]
    key = _keys(cand, tuple(out_shape))
    _, first = np.unique(key, return_index=True)
    return cand[np.sort(first)].astype(np.int32)


def down_rulebook(out_coords: np.ndarray, in_coords: np.ndarray) -> np.ndarray:
    """nbr[k, o] = fine row at 2o - 1 + k or -1."""
    look = _Lookup(in_coords, _extent(in_coords))
    nbr = np.empty((27, len(out_coords)), np.int64)
    for k, off in enumerate(_OFFS):
        q = out_coords.astype(np.int64).copy()
        q[:, 1:] = 2 * q[:, 1:] - 1 + off
        nbr[k] = look(q)
    return nbr


def up_rulebook(fine_coords: np.ndarray, coarse_coords: np.ndarray) -> np.ndarray:
    """nbr[k, i] = coarse row o with 2o - 1 + k = i, or -1 (the strided pairs, roles swapped)."""
    shape = np.asarray(_extent(fine_coords))
    out_shape = tuple((shape - 1) // 2 + 1)
    look = _Lookup(coarse_coords, out_shape)
    nbr = np.empty((27, len(fine_coords)), np.int64)
    for k, off in enumerate(_OFFS):
        q = fine_coords.astype(np.int64).copy()
        num = q[:, 1:] + 1 - off
        q[:, 1:] = num // 2
        r = look(q)
        r[np.any(num % 2 != 0, axis=1)] = -1
        nbr[k] = r
    return nbr


# ------------------------------------------------------------------------------ network ---
def sparse_conv(x: torch.Tensor, nbr: np.ndarray, w: torch.Tensor, n_out: int) -> torch.Tensor:
    """Gather - matmul - accumulate over kernel offsets in ascending k."""
    cout = w.shape[0]
    wk = w.reshape(cout, -1, w.shape[-1])  # [Cout, K, Cin]
    out = torch.zeros(n_out, cout, dtype=x.dtype)
    for k in range(wk.shape[1]):
        rows = nbr[k]
        sel = np.nonzero(rows >= 0)[0]
        if len(sel):
            out[sel] += x[rows[sel]] @ wk[:, k, :].T
    return out


class OracleNet:
    def __init__(self, weights: Dict[str, np.ndarray], dtype=torch.float32, bn_eps: float = 1e-4, fp16: bool = False):
        """fp16 = True restates the half-precision storage mode (BASELINE.json configs[4]; NOT a reference feature -- the
        reference's inference is float32): convolutions whose channel counts are both multiples of 16 use weights
        rounded to half, and every activation tensor with a multiple of 16 channels is rounded to half where the HIP
        path stores it; the arithmetic in between stays in `dtype`."""
        self.w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()
                  if not k.endswith("num_batches_tracked")}
        self.fp16 = fp16
        if fp16:
            for k, v in self.w.items():
                if k.endswith(".weight") and v.ndim == 5 and v.shape[0] % 16 == 0 and v.shape[-1] % 16 == 0 and "_head." not in k:
                    self.w[k] = v.half().to(dtype)
        self.dtype = dtype
        self.eps = bn_eps
        self.depth = 0
        while f"UNet.{'U.' * (self.depth + 1)}Head.sequence.0.weight" in self.w:
            self.depth += 1  # number of encoder levels (3 for the shipped checkpoints)
        self.trace: Dict[str, torch.Tensor] = {}

    # -- layers --------------------------------------------------------------------------
    def q(self, x):
        """Storage rounding of the half-precision mode (levels with a multiple of 16 channels)."""
        return x.half().to(self.dtype) if self.fp16 and x.shape[1] % 16 == 0 else x

    def bn(self, x, prefix):
        w = self.w
        inv = 1.0 / torch.sqrt(w[prefix + ".running_var"] + self.eps)
        return (x - w[prefix + ".running_mean"]) * inv * w[prefix + ".weight"] + w[prefix + ".bias"]

    def pointwise(self, x, key):
        wt = self.w[key]
        return x @ wt.reshape(wt.shape[0], wt.shape[-1]).T

    def res_block(self, x, nbr, prefix, has_identity_conv):
        n = x.shape[0]
        y = sparse_conv(x, nbr, self.w[prefix + ".sequence.0.weight"], n)
        y = self.q(torch.relu(self.bn(y, prefix + ".sequence.1")))
        y = sparse_conv(y, nbr, self.w[prefix + ".sequence.3.weight"], n)
        y = self.bn(y, prefix + ".sequence.4")
        ident = self.q(self.pointwise(x, prefix + ".identity.0.weight")) if has_identity_conv else x
        return self.q(torch.relu(y + ident))

    def ublock(self, x, coords, prefix, level):
        nbr = subm_rulebook(coords)
        self.rulebooks[f"subm{level}"] = nbr
        self.coords[level] = coords
        x = self.res_block(x, nbr, prefix + ".Head", False)
        self.trace[f"head{level}"] = x
        if level == self.depth:
            return x
        skip = x
        coarse = strided_out_coords(coords)
        dn = down_rulebook(coarse, coords)
        self.rulebooks[f"down{level}"] = dn
        z = sparse_conv(x, dn, self.w[prefix + ".Encode.sequence.0.weight"], len(coarse))
        z = self.q(torch.relu(self.bn(z, prefix + ".Encode.sequence.1")))
        self.trace[f"enc{level}"] = z
        z = self.ublock(z, coarse, prefix + ".U", level + 1)
        up = up_rulebook(coords, coarse)
        self.rulebooks[f"up{level}"] = up
        d = sparse_conv(z, up, self.w[prefix + ".Decode.sequence.0.weight"], len(coords))
        d = self.q(torch.relu(self.bn(d, prefix + ".Decode.sequence.1")))
        self.trace[f"dec{level}"] = d
        x = torch.cat((skip, d), dim=1)
        x = self.res_block(x, nbr, prefix + ".Tail", True)
        self.trace[f"tail{level}"] = x
        return x

    def head(self, x, prefix):
        x = torch.relu(self.bn(self.pointwise(x, prefix + ".sequence.0.weight"), prefix + ".sequence.1"))
        x = torch.relu(self.bn(self.pointwise(x, prefix + ".sequence.3.weight"), prefix + ".sequence.4"))
        return self.pointwise(x, prefix + ".sequence.6.weight")

    # -- model.forward(sparse_input) -------------------------------------------------------
    def forward(self, feats: np.ndarray, coords: np.ndarray) -> Dict[str, np.ndarray]:
        self.trace, self.rulebooks, self.coords = {}, {}, {}
        x = torch.as_tensor(np.asarray(feats)).to(self.dtype)
        x = torch.relu(self.bn(self.pointwise(x, "input_conv.sequence.0.weight"), "input_conv.sequence.1"))
        self.trace["input"] = x
        x = self.ublock(x, np.asarray(coords, dtype=np.int32), "UNet", 0)
        radius = self.head(x, "radius_head")
        direction = torch.nn.functional.normalize(self.head(x, "direction_head"))
        class_l = self.head(x, "class_head")
        return {"radius": radius.numpy(), "direction": direction.numpy(), "class_l": class_l.numpy()}


def inference_tail(radius: np.ndarray, direction: np.ndarray, class_l: np.ndarray):
    """model_inference.py:87-88: medial_vector = exp(radius) * direction; class = argmax (first max)."""
    return np.exp(radius) * direction, np.argmax(class_l, axis=1)[:, None]


def load_weights(path) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
