"""Evaluation loop of the training script, forward only (reference smart_tree/model/train.py:61-84).

`eval_epoch` is the reference's validation / test pass: every batch through the network (HIP sparse UNet) and the three losses
(one HIP reduction, model/loss.py).  `train_epoch` is not built: this package has no backward kernels (SURVEY.md section 8f.4
lists the backward pass as out of scope) and says so when called."""
from __future__ import annotations

import torch

from .helper import get_batch
from .tracker import Tracker


@torch.no_grad()
def eval_epoch(data_loader, model, loss_fn, fp16=False, device=None):
    device = device if device is not None else torch.device("cuda")
    tracker = Tracker()
    was_training = getattr(model, "training", False)
    model.eval()
    for sp_input, targets, mask, _ in get_batch(data_loader, device, fp16):
        preds = model.forward(sp_input)
        tracker.update(loss_fn(preds, targets, mask.to(device)))
    if was_training:
        model.train()
    return tracker


def train_epoch(*args, **kwargs):
    raise NotImplementedError("smart_tree_amd has no backward pass: train with the reference, evaluate / infer here")
