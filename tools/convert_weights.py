"""Convert the reference checkpoints' state_dicts into neutral .npz fixtures.

Run ONCE in the build container (the only place /root/reference exists):

    python tools/convert_weights.py

Reads   /root/reference/smart_tree/model/weights/<name>_model_weights.pt   (torch state_dict,
        loaded with weights_only=True -- the pickled full module `<name>_model.pt` is never touched)
Writes  smart_tree_amd/model/weights/<name>.npz   (one float32/int64 array per state_dict key)
        smart_tree_amd/model/weights/SHA256SUMS   (sha256 over tensors in state_dict order)

The .npz files are data fixtures; no reference source is copied.
"""
from __future__ import annotations

import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference/smart_tree/model/weights")
OUT = Path(__file__).resolve().parents[1] / "smart_tree_amd" / "model" / "weights"
NAMES = ["noble-elevator-58", "peach-forest-65"]


def main() -> int:
    OUT.mkdir(parents=True, exist_ok=True)
    sums = []
    for name in NAMES:
        sd = torch.load(REF / f"{name}_model_weights.pt", weights_only=True, map_location="cpu")
        h = hashlib.sha256()
        arrays = {}
        for key, t in sd.items():
            a = t.detach().cpu().numpy()
            h.update(np.ascontiguousarray(a).tobytes())
            arrays[key] = a
        np.savez(OUT / f"{name}.npz", **arrays)
        sums.append(f"{h.hexdigest()}  {name}  ({len(arrays)} tensors)")
        print(sums[-1])
    (OUT / "SHA256SUMS").write_text("\n".join(sums) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
