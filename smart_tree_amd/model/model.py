"""The Smart_Tree network as the shipped checkpoints define it, executed by HIP kernels.

Reference: `Smart_Tree.forward` (smart_tree/model/model.py:77-87) over `SubMConvBlock`, `UBlock`,
`ResBlock`, `EncoderBlock`, `DecoderBlock`, `SparseFC` (smart_tree/model/model_blocks.py).  The
graph is rebuilt from the state_dict's key names and shapes -- the pickled
module `*_model.pt` is never unpickled.  `forward(sparse_input)` keeps the reference contract:
input exposes `.features [N,3]` / `.indices [N,4] int32 (b,z,y,x)`, output is the dict
{"radius" [N,1] (log radius), "direction" [N,3] (unit), "class_l" [N,2] (logits)}.
"""
from __future__ import annotations

import contextlib

from typing import Dict, Mapping

import numpy as np
import torch

from . import sparse_ops as ops
from .. import profiling

BN_EPS = 1e-4  # attribute stored in the checkpoints (model.py:23 would default to 1e-5)


def _conv_weight(t: torch.Tensor) -> torch.Tensor:
    """[Cout, kz, ky, kx, Cin] -> [K, Cin, Cout] (k = (kz*3+ky)*3+kx)."""
    cout, cin = t.shape[0], t.shape[-1]
    return t.reshape(cout, -1, cin).permute(1, 2, 0).contiguous().float()


class _Affine:
    """Eval-mode BatchNorm1d folded to y*scale + shift (computed in float64, stored float32)."""

    def __init__(self, sd: Mapping[str, torch.Tensor], prefix: str, device):
        g = sd[prefix + ".weight"].double()
        b = sd[prefix + ".bias"].double()
        m = sd[prefix + ".running_mean"].double()
        v = sd[prefix + ".running_var"].double()
        s = g / torch.sqrt(v + BN_EPS)
        self.scale = s.float().contiguous().to(device)
        self.shift = (b - m * s).float().contiguous().to(device)


class Smart_Tree:
    def __init__(self, state_dict: Mapping[str, torch.Tensor], device=torch.device("cuda:0"), fp16: bool = False):
        """fp16 (extension, BASELINE.json configs[4]; the reference's inference is float32): the levels whose channel
        count is a multiple of 16 keep their features AND weights in half precision (f16 matrix-core kernel, float32
        accumulation); level 0 (8 channels), the BatchNorm affines and the heads stay float32."""
        sd = {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v.detach().cpu()
              for k, v in state_dict.items()}
        self._state = sd
        self.device = torch.device(device)
        self.depth = 0
        while f"UNet.{'U.' * (self.depth + 1)}Head.sequence.0.weight" in sd:
            self.depth += 1
        self.planes = [sd[f"UNet.{'U.' * l}Head.sequence.0.weight"].shape[0] for l in range(self.depth + 1)]
        self.w: Dict[str, torch.Tensor] = {}
        self.wp: Dict[str, torch.Tensor] = {}  # MFMA operand order for the Cin, Cout % 16 == 0 convolutions
        self.bn: Dict[str, _Affine] = {}
        self.use_mfma = True
        self.fp16 = bool(fp16)
        self.wp16: Dict[str, torch.Tensor] = {}
        self.wq: Dict[str, torch.Tensor] = {}  # three bf16 planes for the split-bf16 kernel (Cin % 32 == 0: the 32- / 64-channel levels)
        self.use_b3 = True
        for key, t in sd.items():
            if key.endswith(".weight") and t.ndim == 5 and "_head." not in key:
                name = key[: -len(".weight")]
                self.w[name] = _conv_weight(t).to(self.device)
                cin_, cout_ = self.w[name].shape[1], self.w[name].shape[2]
                if self.fp16 and not (ops.mfma_eligible(cin_, cout_, 16) or (cin_, cout_) in ((3, 8), (8, 8), (8, 16), (16, 8))):
                    raise ValueError(f"fp16 storage mode has kernels for the shipped widths only (layer {name}: {cin_} -> {cout_})")
                if ops.mfma_eligible(cin_, cout_, 16):  # (other widths: the generic kernel behind st_sparse_conv_fwd)
                    self.wp[name] = ops.mfma_weight(self.w[name])
                    if self.w[name].shape[1] % 32 == 0 or (self.w[name].shape[0] == 27 and ops.b3_eligible(*self.w[name].shape[1:], self.w[name].shape[1])):
                        self.wq[name] = ops.b3_weight(self.w[name])
                    if self.fp16:
                        self.wp16[name] = ops.mfma_weight16_half(self.w[name])
            elif key.endswith(".running_mean") and "_head." not in key:
                p = key[: -len(".running_mean")]
                self.bn[p] = _Affine(sd, p, self.device)
        # the fused head kernel holds the shipped head shapes (planes[0] = 8 -> 8 -> 4 -> 1 / 3 / 2); a model config with other
        # `*_fc_planes` or `unet_planes[0]` (conf/training.yaml:123-127) runs its heads as pointwise convolutions instead
        self.generic_heads = not self._heads_are_standard(sd)
        self.head_params = None if self.generic_heads else self._pack_heads(sd).to(self.device)
        if self.generic_heads:
            self.head_layers = {}
            for name in ("radius_head", "direction_head", "class_head"):
                mat = lambda i: sd[f"{name}.sequence.{i}.weight"].reshape(sd[f"{name}.sequence.{i}.weight"].shape[0], -1).T.contiguous().float()
                self.head_layers[name] = [(mat(0).unsqueeze(0).to(self.device), _Affine(sd, f"{name}.sequence.1", self.device)),
                                          (mat(3).unsqueeze(0).to(self.device), _Affine(sd, f"{name}.sequence.4", self.device)),
                                          (mat(6).unsqueeze(0).to(self.device), None)]
        self.use_bricks = True  # rulebooks from occupancy bricks when the input carries `brick_hint` (sparse.py); else hash tables
        self.spatial_order = True  # run the network on Morton-ordered rows (same values; see features()); traces keep input order
        self.trace = None  # set to a dict to record every block's output (input, head{l}, enc{l}, dec{l}, tail{l}): parity tests

    # nn.Module look-alikes so reference call sites keep working
    def eval(self):
        return self

    def to(self, device):
        return self if torch.device(device) == self.device else Smart_Tree(self._state, device, self.fp16)

    @staticmethod
    def _heads_are_standard(sd) -> bool:
        for name, nout in (("radius_head", 1), ("direction_head", 3), ("class_head", 2)):
            shapes = [tuple(sd[f"{name}.sequence.{i}.weight"].reshape(sd[f"{name}.sequence.{i}.weight"].shape[0], -1).shape) for i in (0, 3, 6)]
            if shapes != [(8, 8), (4, 8), (nout, 4)]:
                return False
        return True

    def _generic_heads(self, x: torch.Tensor, with_tail: bool):
        """SparseFC heads (model_blocks.py:246-285) of any shape: pointwise convolutions with the BatchNorm affine and ReLU in their
        epilogue; direction normalised as the reference's forward does (F.normalize), tail = model_inference.py:87-88."""
        n = x.shape[0]
        outs = []
        for name in ("radius_head", "direction_head", "class_head"):
            h = x
            for w, aff in self.head_layers[name]:
                h = ops.sparse_conv(h, w, None, n, scale=aff.scale if aff else None, shift=aff.shift if aff else None, relu=aff is not None)
            outs.append(h)
        radius, direction, class_l = outs[0], torch.nn.functional.normalize(outs[1]), outs[2]
        if not with_tail:
            return radius, direction, class_l, None, None
        return radius, direction, class_l, torch.exp(radius) * direction, torch.argmax(class_l, dim=1, keepdim=True)

    @staticmethod
    def _pack_heads(sd) -> torch.Tensor:
        """radius / direction / class heads -> the packed block `st_pointwise_mlp_heads` expects."""
        blocks = []
        for name, nout in (("radius_head", 1), ("direction_head", 3), ("class_head", 2)):
            def mat(i):  # [Cout,1,1,1,Cin] -> [Cin, Cout]
                t = sd[f"{name}.sequence.{i}.weight"]
                return t.reshape(t.shape[0], t.shape[-1]).T.contiguous().double()

            def affine(i):
                g, b = sd[f"{name}.sequence.{i}.weight"].double(), sd[f"{name}.sequence.{i}.bias"].double()
                m, v = sd[f"{name}.sequence.{i}.running_mean"].double(), sd[f"{name}.sequence.{i}.running_var"].double()
                s = g / torch.sqrt(v + BN_EPS)
                return s, b - m * s

            s1, t1 = affine(1)
            s2, t2 = affine(4)
            w3 = torch.zeros(12, dtype=torch.float64)
            w3[: 4 * nout] = mat(6).reshape(-1)
            blocks.append(torch.cat([mat(0).reshape(-1), s1, t1, mat(3).reshape(-1), s2, t2, w3]))
        return torch.cat(blocks).float().contiguous()

    # -- building blocks ---------------------------------------------------------------------
    def _conv(self, name, x, nbr, n_out, x1=None, bn=None, residual=None, relu=False, row_order=None):
        a = self.bn[bn] if bn else None
        out_half = self.fp16 and self.w[name].shape[2] % 16 == 0  # half storage on the levels with >= 16 channels
        return ops.sparse_conv(x, self.w[name], nbr, n_out, x1=x1, scale=a.scale if a else None,
                               shift=a.shift if a else None, residual=residual, relu=relu,
                               wp=self.wp.get(name) if self.use_mfma else None, out_half=out_half,
                               wp16=self.wp16.get(name), row_order=row_order,
                               wq=self.wq.get(name) if self.use_mfma and self.use_b3 and not self.fp16 else None)

    def _res_block(self, prefix, x, nbr, x1=None):
        """ResBlock.forward (model_blocks.py:149-156); x1 != None is the Tail on cat(skip, decoded)."""
        n = x.shape[0]
        h = self._conv(prefix + ".sequence.0", x, nbr, n, x1=x1, bn=prefix + ".sequence.1", relu=True)
        ident = self._conv(prefix + ".identity.0", x, None, n, x1=x1) if x1 is not None else x
        return self._conv(prefix + ".sequence.3", h, nbr, n, bn=prefix + ".sequence.4", residual=ident, relu=True)

    def _ublock(self, prefix, x, pyr, level):
        """UBlock.forward (model_blocks.py:224-243)."""
        x = self._res_block(prefix + ".Head", x, pyr.subm[level])
        self._record(f"head{level}", x)
        if level == self.depth:
            return x
        n_fine, n_coarse = x.shape[0], pyr.coords[level + 1].shape[0]
        z = self._conv(prefix + ".Encode.sequence.0", x, pyr.down[level], n_coarse, bn=prefix + ".Encode.sequence.1",
                       relu=True)
        self._record(f"enc{level}", z)
        z = self._ublock(prefix + ".U", z, pyr, level + 1)
        d = self._conv(prefix + ".Decode.sequence.0", z, pyr.up[level], n_fine, bn=prefix + ".Decode.sequence.1",
                       relu=True, row_order=pyr.up_order[level] if pyr.up_order else None)
        self._record(f"dec{level}", d)
        x = self._res_block(prefix + ".Tail", x, pyr.subm[level], x1=d)
        self._record(f"tail{level}", x)
        return x

    def _record(self, name, x):
        if self.trace is not None:
            self.trace[name] = x

    def features(self, sparse_input):
        """input conv + UNet -> [N, planes[0]] features of the finest level, rows in the input's order.
        Internally the voxels are processed in a spatially coherent order (`spatial_order`, on by default): every output row
        is computed on its own from the rows its neighbour table names, so the values do not depend on the row order, but
        gathers hit rows that are close in memory and the 16 rows of a matrix-core tile share their live offsets."""
        coords = sparse_input.indices.contiguous()
        feats = sparse_input.features.contiguous().float()
        blk_seg, n_seg = getattr(sparse_input, "blk_seg", None), getattr(sparse_input, "n_seg", 1)
        reorder = self.spatial_order and self.trace is None and coords.shape[0] > 1
        hint = getattr(sparse_input, "brick_hint", None)
        bricks = ops.brick_pyramid(coords, self.depth, hint[0], hint[1], blk_seg, n_seg) if reorder and hint and self.use_bricks else None
        if bricks is not None:  # occupancy bricks: the order and every table from one call, no hash probes, one read-back
            pyr, order = bricks
            feats = ops.move_rows(feats, order)
        else:
            order = ops.spatial_order(coords) if reorder else None
            if order is not None:
                coords, feats = ops.move_rows(coords, order), ops.move_rows(feats, order)
            pyr = ops.build_pyramid(coords, self.depth, blk_seg, n_seg)
        gate = getattr(self, "conv_gate", None)  # optional context-manager factory around the convolution launches (no host
        #                                            synchronisation inside): a caller with several batches in flight can keep
        #                                            their conv sequences from sharing the chip (bench.py)
        with (gate() if gate is not None else contextlib.nullcontext()), profiling.kernel_family("k_sparse_conv* (one forward pass)"):
            x = self._conv("input_conv.sequence.0", feats, None, feats.shape[0], bn="input_conv.sequence.1", relu=True)
            self._record("input", x)
            x = self._ublock("UNet", x, pyr, 0)
        if order is not None:
            x = ops.move_rows(x, order, scatter=True)
        return x

    def forward(self, sparse_input) -> Dict[str, torch.Tensor]:
        x = self.features(sparse_input)
        radius, direction, class_l, _, _ = self._generic_heads(x, False) if self.generic_heads else ops.mlp_heads(x, self.head_params)
        return {"radius": radius, "direction": direction, "class_l": class_l}

    __call__ = forward

    def forward_fused_tail(self, sparse_input):
        """forward() plus ModelInference's exp(radius)*direction and argmax in the same kernel."""
        x = self.features(sparse_input)
        return self._generic_heads(x, True) if self.generic_heads else ops.mlp_heads(x, self.head_params, with_tail=True)
