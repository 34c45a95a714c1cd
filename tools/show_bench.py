"""Developer aid: the fields of a bench.py JSON line that matter when comparing runs.  python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value {d['value'] / 1e6:.1f} M points/s, {d['ms_per_step']:.3f} ms per step (steps {d['steps']}, warm-up {d['warmup']}); "
      f"passes {d['passes']['ms_per_step']}")
print(f"incl. host upload {d['value_incl_host_upload'] / 1e6:.1f} M points/s; passes {d['incl_host_upload']['ms_per_step']}")
c = d["config"]
print("config:", {k: c.get(k) for k in ("distinct_clouds_per_rank", "clouds_per_launch_set", "batches_in_timed_region", "batches_in_flight_per_gpu",
                                        "cloud_generation_s", "warmup_steps_run", "host_cpu_cores_busy_in_timed_region")})
r = d["roofline"] or {}
print("roofline:", {k: r.get(k) for k in ("kernel", "bound", "frac", "achieved", "avg_us", "launches", "total_ms", "dominant_by_chip_time", "traffic",
                                          "algorithmic_bytes_per_launch")})
print("gather/scatter:", d.get("roofline_gather_scatter"))
s = d.get("single_cloud")
if s:
    print(f"single cloud: {s['ms']} ms (min {s['ms_min']}, max {s['ms_max']}, {s['calls']} calls) = {s['points_per_s'] / 1e6:.1f} M points/s")
    print("   stages:", s["stage_ms"])
    print("   conv family:", s["roofline_gather_scatter"], "largest:", s["largest_kernel"])
print("stage_ms:", d.get("stage_ms"))
if d.get("cpu_baseline"):
    print("cpu baseline:", d["cpu_baseline"]["value"], "points/s on", d["cpu_baseline"]["cores"], "cores;", "parity_in_run", d.get("parity_in_run"),
          "branches compared", d.get("branches_compared"))
if d.get("roofline_solo"):
    rs = d["roofline_solo"]
    print("solo:", {k: rs.get(k) for k in ("kernel", "frac", "avg_us")}, "gather_gemm", (rs.get("gather_gemm") or {}).get("hbm_frac"))
for k, v in (d.get("other_configs") or {}).items():
    print("other:", k[:40], v.get("ms_per_cloud"), v.get("gather_gemm", {}).get("hbm_frac"))
