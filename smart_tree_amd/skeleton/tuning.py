"""Per-call tuning of the skeleton stage's kernels (test hook + sweep aid; the defaults are the product settings).

The library keeps NO process-global knobs: `st_skeleton_components_seg` takes an optional array of 24 int64 (entry =
"default" or a value, codes in csrc/skeleton.hip "Tuning of one call"), the neighbour searches an optional cap multiplier of
their grid cell.  This module holds what the CALLING THREAD wants passed: `with tuning.override({5: -1, 14: 1}): ...` in
tests/test_skeleton.py forces every claim strategy / SSSP form; `ST_SKELETON_PARAMS="3=24,9=3"` seeds the defaults of a
process for the sweep scripts under tools/.  Results never depend on any of it (that is what the tests check)."""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading

DEFAULT = -(1 << 63)  # ST_TUNE_DEFAULT
KNN_CELL_MEAN_MULT = 100  # pseudo code (outside the library's 0..23): hundredths of the mean bound that caps the search-grid cell (-> st_knn_radius_seg)
TICKS = 15
HELP_LIFETIME_US, HELP_TIMEOUT_US, HELP_ANNOUNCE_US = 16, 17, 18  # time-outs of the helper protocol (tests/test_helpers.py)
ENTRIES = 24

_local = threading.local()


def _env_defaults() -> dict:
    out = {}
    for kv in filter(None, os.environ.get("ST_SKELETON_PARAMS", "").split(",")):
        which, value = kv.split("=")
        out[int(which)] = int(value)
    return out


_ENV = _env_defaults()


def current() -> dict:
    return {**_ENV, **getattr(_local, "values", {})}


@contextlib.contextmanager
def override(values: dict):
    """Knob code -> value for every skeleton / search call of this thread inside the block."""
    before = getattr(_local, "values", {})
    _local.values = {**before, **{int(k): int(v) for k, v in values.items()}}
    try:
        yield
    finally:
        _local.values = before


def skeleton_array():
    """The `tuning` argument of st_skeleton_components_seg (None = all defaults)."""
    cur = {k: v for k, v in current().items() if 0 <= k < ENTRIES}
    if not cur:
        return None
    arr = (ctypes.c_int64 * ENTRIES)(*([DEFAULT] * ENTRIES))
    for k, v in cur.items():
        arr[k] = v
    return arr


def knn_cell_mean_mult() -> float:
    """The `cell_mean_mult` argument of st_knn_radius_seg / st_radius_count_seg (-1 = library default)."""
    v = current().get(KNN_CELL_MEAN_MULT)
    return -1.0 if v is None else v / 100.0


COUNT_CELL_MEAN_MULT = 101  # pseudo code: the same cap for outlier_removal's counting search alone (falls back to KNN_CELL_MEAN_MULT)


def count_cell_mean_mult() -> float:
    v = current().get(COUNT_CELL_MEAN_MULT)
    return knn_cell_mean_mult() if v is None else v / 100.0
