"""Evaluation-side losses, forward only (reference smart_tree/model/loss.py:7-97).

`compute_loss` has the reference's signature.  When it is handed THIS module's loss functions -- `L1Loss`,
`cosine_similarity_loss`, `focal_loss` / `dice_loss`, the combination the reference trains with -- the masks, the logarithm of the
target radius and the three reductions run as ONE HIP pass over the voxels (`st_loss_forward`, csrc/loss.hip) instead of a
dozen torch kernels and three boolean compactions; with any other callables it does what the reference does, step by step, and
calls them.  Used on their own, the functions below reduce whatever tensors they are given through the same kernel.
There is no backward pass in this package: the returned tensors carry no graph.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _lib

_OUT = {"radius": 0, "direction": 1, "focal": 2, "dice": 3}


def _forward(radius, direction, class_l, targets, mask, vector_class, target_radius_log):
    L = _lib.lib()
    dev = radius.device
    f = lambda t: t.detach().contiguous().float()
    radius, direction, class_l, targets = f(radius).view(-1), f(direction), f(class_l), f(targets)
    n = radius.shape[0]
    if direction.shape != (n, 3) or class_l.dim() != 2 or class_l.shape[0] != n or targets.shape[0] != n:
        raise ValueError(f"loss: shapes {tuple(radius.shape)}, {tuple(direction.shape)}, {tuple(class_l.shape)}, {tuple(targets.shape)}")
    m = None
    if mask is not None:
        if mask.dtype != torch.bool or mask.shape[0] != n:
            raise ValueError("loss: mask must be a boolean tensor with one entry per row")
        m = mask.to(dev).contiguous().view(torch.uint8)
    out = (ctypes.c_double * 8)()
    ws = _lib.workspace(L.st_loss_workspace_bytes(), dev)
    _lib.check(L.st_loss_forward(_lib.ptr(radius), _lib.ptr(direction), _lib.ptr(class_l), class_l.shape[1], _lib.ptr(targets),
                                 targets.shape[1], _lib.ptr(m), n, -1 if vector_class is None else int(vector_class),
                                 1 if target_radius_log else 0, out, _lib.ptr(ws), ws.numel(), _lib.stream(dev)))
    return [torch.tensor(v, dtype=torch.float32, device=dev) for v in out[:4]]


def L1Loss(outputs, targets):
    """loss.py:54-56: mean |outputs - targets|."""
    o = outputs.reshape(-1)
    n = o.shape[0]
    t5 = torch.zeros((n, 5), dtype=torch.float32, device=o.device)
    t5[:, 0] = targets.reshape(-1)
    z = torch.zeros((n, 3), dtype=torch.float32, device=o.device)
    return _forward(o, z, z[:, :1], t5, None, None, False)[0]


def cosine_similarity_loss(outputs, targets):
    """loss.py:59-61: mean 1 - cos(outputs, targets), torch's CosineSimilarity (each norm clamped at 1e-8)."""
    n = outputs.shape[0]
    t5 = torch.zeros((n, 5), dtype=torch.float32, device=outputs.device)
    t5[:, 1:4] = targets
    z = torch.zeros((n, 1), dtype=torch.float32, device=outputs.device)
    return _forward(z.view(-1), outputs, z, t5, None, None, False)[1]


def _class_loss(outputs, targets, which):
    if outputs.dim() > 2:  # N,C,H,W -> N*H*W,C (loss.py:86-89)
        outputs = outputs.reshape(outputs.size(0), outputs.size(1), -1).transpose(1, 2).reshape(-1, outputs.size(1))
    n = outputs.shape[0]
    t5 = torch.zeros((n, 5), dtype=torch.float32, device=outputs.device)
    t5[:, 4] = targets.reshape(-1).float()
    z = torch.zeros((n, 3), dtype=torch.float32, device=outputs.device)
    return _forward(z[:, 0], z, outputs, t5, None, None, False)[which]


def focal_loss(outputs, targets):
    """loss.py:81-97, gamma = 2: mean -(1 - p_t)^2 log p_t."""
    return _class_loss(outputs, targets, _OUT["focal"])


def dice_loss(outputs, targets):
    """loss.py:64-78: 1 - (2 sum(softmax * onehot) + 1) / (sum softmax + sum onehot + 1) (one_hot over the logits' classes)."""
    return _class_loss(outputs, targets, _OUT["dice"])


def nll_loss(outputs, targets):
    """loss.py:100-103: the reference returns a constant zero (the weighted NLL below its `return` is dead code)."""
    return torch.tensor([0], device=outputs.device)


_FUSED_CLASS = {focal_loss: _OUT["focal"], dice_loss: _OUT["dice"]}


def compute_loss(preds, targets, mask=None, radius_loss_fn=None, direction_loss_fn=None, class_loss_fn=None,
                 target_radius_log=True, vector_class=None):
    """loss.py:7-51.  preds: {"radius" [n,1], "direction" [n,3], "class_l" [n,C]}; targets [n,5] = radius, direction, class.
    The fused dice term one-hot encodes over the C classes of the logits; the reference's `F.one_hot(targets)` infers the width
    from the largest id PRESENT and raises a shape mismatch for a batch that lacks class C-1 -- here such a batch is evaluated."""
    if radius_loss_fn is L1Loss and direction_loss_fn is cosine_similarity_loss and class_loss_fn in _FUSED_CLASS:
        out = _forward(preds["radius"], preds["direction"], preds["class_l"], targets, mask, vector_class, target_radius_log)
        return {"radius": out[0], "direction": out[1], "class_l": out[_FUSED_CLASS[class_loss_fn]]}
    # foreign loss functions: the reference's own sequence of selections
    radius, direction, class_l = preds["radius"], preds["direction"], preds["class_l"]
    t_class, t_dir, t_rad = targets[:, [-1]].long(), targets[:, 1:-1], targets[:, [0]]
    if mask is not None:
        radius, direction, class_l = radius[mask], direction[mask], class_l[mask]
        t_rad, t_dir, t_class = t_rad[mask], t_dir[mask], t_class[mask]
    if vector_class is not None:
        vm = (t_class == vector_class).view(-1)
        radius, direction, t_rad, t_dir = radius[vm], direction[vm], t_rad[vm], t_dir[vm]
    if target_radius_log:
        t_rad = torch.log(t_rad)
    return {"radius": radius_loss_fn(radius.view(-1), t_rad.view(-1)), "direction": direction_loss_fn(direction, t_dir),
            "class_l": class_loss_fn(class_l, t_class)}


@torch.no_grad()
def evaluate_losses(batches, model, loss_fn, device=None) -> dict:
    """Forward-only evaluation of collated batches (`model.sparse.batch_collate` items: ((inputs, targets), coords, loss_mask,
    names)): every batch through the network and `loss_fn(preds, targets, mask)`; returns the mean of each loss term over the
    batches plus their sum under "total".  This is the part of the reference's validation pass (train.py:61-84) that is a
    forward computation; the optimiser loop, the logger and the backward pass are out of scope."""
    from .sparse import sparse_from_batch

    device = torch.device(device) if device is not None else torch.device("cuda")
    sums, count = {}, 0
    for (feats, target_feats), coords, mask, _ in batches:
        preds = model.forward(sparse_from_batch(feats, coords, device=device))
        for k, v in loss_fn(preds, target_feats.to(device), mask.to(device)).items():
            sums[k] = sums.get(k, 0.0) + float(v)
        count += 1
    out = {k: v / max(count, 1) for k, v in sums.items()}
    out["total"] = sum(out.values())
    return out
