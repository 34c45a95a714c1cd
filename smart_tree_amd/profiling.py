"""Opt-in stage / kernel timers (HIP events on the current stream) used by bench.py.

Disabled by default: `stage()` / `kernel()` are no-ops unless `enable(True)` was called, so the
product path pays nothing.  Events are recorded on torch's current HIP stream -- the stream every
kernel of this package is launched on (`_lib.stream()`).
"""
from __future__ import annotations

from collections import defaultdict
from contextlib import contextmanager

import torch

_enabled = False
_stages = defaultdict(list)  # name -> [(start, end)]
_kernels = defaultdict(list)  # name -> [(start, end, algorithmic_bytes)]


def enable(flag: bool) -> None:
    global _enabled
    _enabled = bool(flag) and torch.cuda.is_available()
    if flag:
        _stages.clear()
        _kernels.clear()
        _external.clear()


def enabled() -> bool:
    return _enabled


@contextmanager
def stage(name: str):
    if not _enabled:
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _stages[name].append((a, b))


@contextmanager
def kernel(name: str, algorithmic_bytes: float):
    if not _enabled:
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _kernels[name].append((a, b, algorithmic_bytes))  # a number, or a thunk evaluated after the timed region


_external = defaultdict(list)  # name -> [(total_ms, launches, bytes_thunk)] measured by the library itself


def add_kernel_time(name: str, total_ms: float, launches: int, total_bytes) -> None:
    """Kernel time the C library measured with its own HIP events (launch loops that live in C)."""
    if _enabled:
        _external[name].append((float(total_ms), int(launches), total_bytes))


def stage_ms(steps: int):
    torch.cuda.synchronize()
    return {k: round(sum(a.elapsed_time(b) for a, b in v) / max(steps, 1), 3) for k, v in _stages.items()}


def kernel_table():
    torch.cuda.synchronize()
    rows = {}
    for name, recs in _kernels.items():
        ms = [a.elapsed_time(b) for a, b, _ in recs]
        nbytes = [float(r[2]() if callable(r[2]) else r[2]) for r in recs]
        rows[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1e3 * sum(ms) / len(ms),
                      "bytes_per_launch": sum(nbytes) / len(recs)}
    for name, recs in _external.items():
        launches = sum(r[1] for r in recs)
        total = sum(r[0] for r in recs)
        nbytes = sum(float(r[2]() if callable(r[2]) else r[2]) for r in recs)
        rows[name] = {"launches": launches, "total_ms": total, "avg_us": 1e3 * total / max(launches, 1),
                      "bytes_per_launch": nbytes / max(launches, 1)}
    return rows


def roofline(peak_gbs: float):
    """Roofline entry of the kernel class with the largest total time in the timed region."""
    rows = kernel_table()
    if not rows:
        return None
    name = max(rows, key=lambda k: rows[k]["total_ms"])
    r = rows[name]
    achieved = r["bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
    return {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": peak_gbs, "unit": "GB/s",
            "frac": achieved / peak_gbs, "traffic": None, "launches": r["launches"], "avg_us": r["avg_us"],
            "algorithmic_bytes_per_launch": r["bytes_per_launch"],
            "all_kernels": {k: {"total_ms": round(v["total_ms"], 3), "launches": v["launches"]} for k, v in rows.items()}}
