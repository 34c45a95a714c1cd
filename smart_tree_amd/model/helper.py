"""Batch plumbing of the evaluation loop (reference smart_tree/model/helper.py:13-33)."""
from __future__ import annotations

import torch

from .sparse import sparse_from_batch


def get_batch(dataloader, device, fp_16=False):
    """helper.py:13-33: (sparse input, targets on the device, loss mask, file names) for every collated batch.  `fp_16`
    rounds features and targets to half precision as the reference does (the coordinates stay integers here)."""
    for (feats, target_feats), coords, mask, filenames in dataloader:
        if fp_16:
            feats, target_feats = feats.half().float(), target_feats.half().float()
        yield sparse_from_batch(feats, coords, device=device), target_feats.to(device), mask, filenames
