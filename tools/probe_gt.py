"""GPU box: the skeleton stage on configs[1]'s trees with their GROUND-TRUTH medial vectors (bench.representative_inputs (a)):
    python tools/probe_gt.py [n_set=20] [params "k=v,..."] [parity 0/1] [training_scale 0/1]
Prints one JSON object: single call + one launch set (ms, graph size, SSSP rounds, selection rounds / phases, helpers)."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
n_set = int(sys.argv[1]) if len(sys.argv) > 1 else 20
params = sys.argv[2] if len(sys.argv) > 2 else ""
parity = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
training = (sys.argv[4] if len(sys.argv) > 4 else "0") == "1"

import bench  # noqa: E402

import ctypes as _ct, os as _os
from smart_tree_amd import _lib as _l
if _os.environ.get("ST_PROBE_LIB"):  # A/B aid: another build of the library (e.g. the previous revision's .so kept under _ab/)
    _l._LIB = _l.declare(_ct.CDLL(_os.environ["ST_PROBE_LIB"]))

host_xyz, host_mv = bench.generate_clouds(bench.N_POINTS, list(range(n_set)), min(n_set, bench.usable_cores()), n_mv=n_set)
import torch  # noqa: E402

from smart_tree_amd.skeleton import tuning  # noqa: E402

knobs = {}
for kv in filter(None, params.split(",")):
    k, v = kv.split("=")
    knobs[int(k)] = int(v)
with tuning.override(knobs):
    out = bench.representative_inputs(torch.device("cuda:0"), host_xyz, host_mv, n_set=n_set, check_parity=parity, training_scale=training)
print(json.dumps({"params": params, **out}))
