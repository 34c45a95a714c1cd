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


_generation = 0  # bumped by every enable(True): caches of per-launch facts (pair counts keyed by tensor id) start afresh


def generation() -> int:
    return _generation


def enable(flag: bool) -> None:
    global _enabled, _generation
    _enabled = bool(flag) and torch.cuda.is_available()
    if flag:
        _generation += 1
        _stages.clear()
        _kernels.clear()
        _external.clear()
        _families.clear()


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
def kernel(name: str, algorithmic_bytes, flops=0):
    if not _enabled:
        yield
        return
    pending = getattr(_tls, "pending", None)
    if pending is not None:  # inside a kernel_family block in family mode: counted, not bracketed
        pending.append((algorithmic_bytes, flops))
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _kernels[name].append((a, b, algorithmic_bytes, flops))  # numbers, or thunks evaluated after the timed region


_external = defaultdict(list)  # name -> [(total_ms, launches, bytes_thunk)] measured by the library itself

# Family mode: a pair of events around every launch costs the GPU ~10 us between two kernels of a back-to-back sequence
# (measured: 26 convolutions of a forward pass = 0.28 ms per launch set, 3 % of the chip-filling phase of an 8-cloud batch).
# With family_mode(True) the launches inside a `kernel_family(...)` block are NOT bracketed one by one: the block is, once, and
# every `kernel(...)` inside it only hands over its byte / flop counts.  The block's duration is then the sum of its launches'
# durations plus their (sub-microsecond) hand-overs -- what the roofline prices a kernel family with anyway.
_family = False
_families = defaultdict(list)  # name -> [(start, end, [(bytes, flops), ...])]
_tls = __import__("threading").local()  # the open block of THIS host thread (bench.py runs one thread per stream)


def family_mode(flag: bool) -> None:
    global _family
    _family = bool(flag)


@contextmanager
def kernel_family(name: str):
    if not (_enabled and _family):
        yield
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _tls.pending = []
    try:
        yield
    finally:
        b.record()
        items, _tls.pending = _tls.pending, None
        _families[name].append((a, b, items))


_cus = {}


def compute_units(dev) -> int:
    """Compute units of the device (for the chip share of a launch that cannot fill it)."""
    key = str(dev)
    if key not in _cus:
        try:
            idx = dev.index if getattr(dev, "index", None) is not None else torch.cuda.current_device()
            _cus[key] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
        except Exception:  # noqa: BLE001 -- a diagnostic must not take the run down
            _cus[key] = 256  # MI355X
    return _cus[key]


def add_kernel_time(name: str, total_ms: float, launches: int, total_bytes, chip_share: float = 1.0) -> None:
    """Kernel time the C library measured with its own HIP events (launch loops that live in C).  chip_share: the fraction of
    the 256 compute units the launch's grid can occupy (a one-workgroup-per-tree kernel on a batch of 20 trees: 20 / 256)."""
    if _enabled:
        _external[name].append((float(total_ms), int(launches), total_bytes, min(1.0, max(float(chip_share), 0.0))))


def settle() -> None:
    """Evaluate every pending byte / flop thunk NOW and keep the numbers only.  A thunk pins what it counts (a conv's neighbour
    table, a selection's result arrays); a timed region of several passes calls this between passes -- outside the timed
    brackets -- so that the tables of all its batches do not pile up in HBM until the tables are read."""
    global _generation
    torch.cuda.synchronize()
    val = lambda t: float(t() if callable(t) else t)
    for recs in _kernels.values():
        for i, r in enumerate(recs):
            recs[i] = (r[0], r[1], val(r[2]), val(r[3]))
    for recs in _families.values():
        for i, r in enumerate(recs):
            recs[i] = (r[0], r[1], [(val(b), val(f)) for b, f in r[2]])
    for recs in _external.values():
        for i, r in enumerate(recs):
            recs[i] = (r[0], r[1], val(r[2]), r[3])
    _generation += 1  # per-tensor caches keyed by (id, generation) must not outlive the tensors released here


def stage_ms(steps: int):
    torch.cuda.synchronize()
    return {k: round(sum(a.elapsed_time(b) for a, b in v) / max(steps, 1), 3) for k, v in _stages.items()}


def kernel_table():
    torch.cuda.synchronize()
    rows = {}
    for name, recs in _kernels.items():
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        nbytes = [float(r[2]() if callable(r[2]) else r[2]) for r in recs]
        nflops = [float(r[3]() if callable(r[3]) else r[3]) for r in recs]
        rows[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1e3 * sum(ms) / len(ms),
                      "bytes_per_launch": sum(nbytes) / len(recs), "flops_per_launch": sum(nflops) / len(recs)}
    val = lambda t: float(t() if callable(t) else t)
    for name, recs in _families.items():
        launches = sum(len(r[2]) for r in recs)
        total = sum(r[0].elapsed_time(r[1]) for r in recs)
        rows[name] = {"launches": launches, "total_ms": total, "avg_us": 1e3 * total / max(launches, 1),
                      "bytes_per_launch": sum(val(i[0]) for r in recs for i in r[2]) / max(launches, 1),
                      "flops_per_launch": sum(val(i[1]) for r in recs for i in r[2]) / max(launches, 1)}
    for name, recs in _external.items():
        launches = sum(r[1] for r in recs)
        total = sum(r[0] for r in recs)
        nbytes = sum(float(r[2]() if callable(r[2]) else r[2]) for r in recs)
        rows[name] = {"launches": launches, "total_ms": total, "avg_us": 1e3 * total / max(launches, 1),
                      "bytes_per_launch": nbytes / max(launches, 1), "flops_per_launch": 0.0,
                      "chip_ms": sum(r[0] * r[3] for r in recs)}
    return rows


def _pmc_traffic(kernel_name: str):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary (profiles/*_pmc_summary.csv:
    FETCH_SIZE + WRITE_SIZE, KB per dispatch), or None.  PMC counters cannot be read from inside the process.
    A name ending in '*' is a kernel family (every instantiation of a template): launch-weighted mean over its rows."""
    import csv
    from pathlib import Path

    files = sorted((Path(__file__).resolve().parents[1] / "profiles").glob("*_pmc_summary.csv"))
    if not files:
        return None
    family = kernel_name.endswith("*")
    stem = kernel_name.rstrip("*").split("<")[0]
    total, launches = 0.0, 0
    for row in csv.DictReader(open(files[-1])):
        base = row["kernel"].split("(")[0].split("<")[0].strip().split(" ")[-1]  # "void k_x<..>(..)" -> "k_x"
        if (base.startswith(stem) if family else base == stem):
            n = int(row["launches"] or 0)
            # gfx950 correction (MI355X_MICROARCH.md "HBM" + the calibration it asks for, profiles/r02_pmc_calibration.txt):
            # FETCH_SIZE tallies every 128-byte line at 64 B, for streams and gathers alike -> x2; WRITE_SIZE is exact
            total += n * (2.0 * float(row["FETCH_SIZE_KB_per_launch"] or 0) + float(row["WRITE_SIZE_KB_per_launch"] or 0)) * 1024.0
            launches += n
    return total / launches if launches else None


PMC_CLOUDS_PER_LAUNCH = 16  # batch size of the committed PMC passes (tools/pmc_select.sh, tools/collect_r2.sh: --streams 1 --steps 16)


def _scaled_traffic(name: str, clouds_per_launch: int):
    t = _pmc_traffic(name)
    if t is None or not clouds_per_launch:
        return t
    return t * clouds_per_launch / PMC_CLOUDS_PER_LAUNCH


LATENCY_BOUND = ("k_sk_select", "k_sk_sssp")  # one workgroup (cluster) per tree / level-synchronous frontier: priced against the HBM peak all the same


def roofline(peak_gbs: float, peak_f32_tflops: float = 157.3, clouds_per_launch: int = 0):
    """Roofline entry of the kernel class with the largest total time in the timed region, plus (as `gather_gemm`) the
    AGGREGATE over every sparse-conv launch -- sum of algorithmic bytes / sum of kernel time: the gather / rule-GEMM /
    scatter the path is named for -- with the per-class table in `all_kernels`."""
    rows = kernel_table()
    if not rows:
        return None
    # The sparse convolution is ONE kernel family (two templates, k_sparse_conv / k_sparse_conv_mfma, instantiated per channel
    # pair): its instantiations are priced together -- sum of algorithmic bytes / sum of launch time -- and compete as one
    # entry for "the kernel with the largest total time"; every instantiation is still listed in `all_kernels`.
    CONV = "k_sparse_conv*"
    cands = {k: v for k, v in rows.items() if not k.startswith("k_sparse_conv")}
    fam = [v for k, v in rows.items() if k.startswith("k_sparse_conv")]
    if fam:
        n = sum(v["launches"] for v in fam)
        tot = sum(v["total_ms"] for v in fam)
        cands[CONV] = {"launches": n, "total_ms": tot, "avg_us": 1e3 * tot / n,
                       "bytes_per_launch": sum(v["bytes_per_launch"] * v["launches"] for v in fam) / n,
                       "flops_per_launch": sum(v["flops_per_launch"] * v["launches"] for v in fam) / n}
    # "largest total time" = plain summed launch duration in the timed region (the rule of rounds 1-2; round 3 weighted a
    # launch by the share of the chip its grid can occupy, which moved the entry from the latency-bound branch selection to the
    # convolutions -- that ranking is kept as `dominant_by_chip_time` for the record, it no longer picks the entry).
    name = max(cands, key=lambda k: cands[k]["total_ms"])
    by_chip = max(cands, key=lambda k: cands[k].get("chip_ms", cands[k]["total_ms"]))
    r = cands[name]
    achieved = r["bytes_per_launch"] / (r["avg_us"] * 1e-6) / 1e9
    latency_bound = name in LATENCY_BOUND
    notes = {
        "k_sk_select": "latency-bound branch selection (priced against the HBM peak all the same): one workgroup per tree "
                       "component runs speculative rounds (each of its 16 wavefronts walks one candidate tip; ~25 us of dependent "
                       "accesses and barriers per round, 3-4 branches accepted per round); bytes = path*24 + claimed*16 + "
                       "8*vertices per tree, divided over its launches; a batch of clouds runs its components side by side",
        CONV: "gather / rule-GEMM / scatter: every instantiation of the two sparse-conv templates.  In the timed region the "
              "convolutions of a forward pass are bracketed ONCE (HIP events before the first and after the last launch: a pair "
              "per launch costs ~10 us between kernels that otherwise run back to back), avg_us = bracket / launches; launches "
              "of the batches in flight share the chip, so a bracket in the timed region is longer than solo -- roofline_solo "
              "has the same entry for one batch alone on the GPU with every launch bracketed (per-class table in all_kernels)",
    }
    out = {"kernel": name, "bound": "latency" if latency_bound else "hbm", "achieved": achieved, "peak": peak_gbs, "unit": "GB/s",
           "frac": achieved / peak_gbs, "traffic": _scaled_traffic(name, clouds_per_launch), "launches": r["launches"], "avg_us": r["avg_us"],
           "traffic_note": f"HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, two rocprofv3 PMC passes, profiles/*_pmc_summary.csv) "
                           f"measured at {PMC_CLOUDS_PER_LAUNCH} clouds per launch set and scaled linearly to this run's "
                           f"{clouds_per_launch or PMC_CLOUDS_PER_LAUNCH}; gfx950 correction: 2 x FETCH_SIZE + WRITE_SIZE (FETCH_SIZE counts a "
                           "128-byte line as 64 B for streams and gathers alike, WRITE_SIZE is exact: profiles/r02_pmc_calibration.txt); "
                           "a kernel family = launch-weighted mean over its instantiations",
           "algorithmic_bytes_per_launch": r["bytes_per_launch"],
           "dominant_by": "summed launch duration in the timed region (a kernel family = one entry)",
           "dominant_by_chip_time": by_chip,
           "total_ms": r["total_ms"],
           "note": notes.get(name, "")}
    if "k_sk_select" in rows and name != "k_sk_select":  # the (latency-bound) runner-up, for the record
        q = rows["k_sk_select"]
        a = q["bytes_per_launch"] / (q["avg_us"] * 1e-6) / 1e9
        out["branch_selection"] = {"kernel": "k_sk_select", "bound": "latency", "achieved": a, "frac": a / peak_gbs,
                                   "launches": q["launches"], "avg_us": q["avg_us"], "total_ms": q["total_ms"],
                                   "traffic": _scaled_traffic("k_sk_select", clouds_per_launch),
                                   "algorithmic_bytes_per_launch": q["bytes_per_launch"], "note": notes["k_sk_select"]}
    convs = {k: v for k, v in rows.items() if k.startswith("k_sparse_conv")}
    if convs:
        tot_ms = sum(v["total_ms"] for v in convs.values())
        tot_b = sum(v["bytes_per_launch"] * v["launches"] for v in convs.values())
        tot_f = sum(v["flops_per_launch"] * v["launches"] for v in convs.values())
        gbs = tot_b / (tot_ms * 1e-3) / 1e9
        tfs = tot_f / (tot_ms * 1e-3) / 1e12
        cname = max(convs, key=lambda k: convs[k]["total_ms"])
        out["gather_gemm"] = {"kernels": "all sparse-conv launches (sum of bytes / sum of time)", "bound": "hbm",
                              "launches": sum(v["launches"] for v in convs.values()), "total_ms": tot_ms,
                              "algorithmic_bytes_total": tot_b, "achieved_GBps": gbs, "peak": peak_gbs, "hbm_frac": gbs / peak_gbs,
                              "useful_TFLOPs": tfs, "f32_matrix_frac": tfs / peak_f32_tflops, "largest_class": cname}
    out["all_kernels"] = {k: {"total_ms": round(v["total_ms"], 3), "launches": v["launches"],
                              "GBps": round(v["bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9, 1)} for k, v in rows.items()}
    return out
