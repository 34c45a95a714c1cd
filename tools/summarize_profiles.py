"""Copy the rocprofv3 outputs merged into gpurun_out/ into tracked, judged summaries under profiles/.

    python tools/summarize_profiles.py r01
"""
import collections, csv, glob, json, os, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)

for name in (f"{tag}_bench.json", f"{tag}_bench_steps20.json"):  # the default command's line and the driver's (--steps 20)
    bench = ROOT / "gpurun_out" / name
    if bench.exists():
        line = [l for l in bench.read_text().splitlines() if l.startswith("{")][-1]
        (out / name).write_text(json.dumps(json.loads(line), indent=1) + "\n")

stats = sorted(glob.glob(str(ROOT / "gpurun_out" / f"prof_{tag}" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], out / f"{tag}_kernel_stats.csv")
stats20 = sorted(glob.glob(str(ROOT / "gpurun_out" / f"prof_{tag}_steps20" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
if stats20:  # the DRIVER's command (--steps 20)
    shutil.copy(stats20[-1], out / f"{tag}_kernel_stats_steps20.csv")


def counter(pattern, name):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    if files:
        for r in csv.DictReader(open(files[-1])):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fe = counter(str(ROOT / "gpurun_out" / "pmc_fetch" / "*" / "*counter_collection.csv"), "FETCH_SIZE")
wr = counter(str(ROOT / "gpurun_out" / "pmc_write" / "*" / "*counter_collection.csv"), "WRITE_SIZE")
if fe:
    with open(out / f"{tag}_pmc_summary.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"])
        for k in sorted(fe, key=lambda k: -sum(fe[k])):
            ws = wr.get(k, [])
            w.writerow([k, len(fe[k]), round(sum(fe[k]) / len(fe[k]), 2), round(sum(ws) / len(ws), 2) if ws else ""])


def body(name):
    f = ROOT / "gpurun_out" / name
    if not f.exists():
        return None
    return "".join(l for l in f.read_text().splitlines(keepends=True) if "amdgpu.ids" not in l)


CONV_HDR = ("# per-layer time of the sparse-conv kernels on that cloud's rulebooks, 20 launches each, random features / weights;\n"
            "# the inverse (\"up\") convs run in coordinate-parity order, as in the model.\n"
            "# VALU = float32 vector kernel, mfma = float32 matrix-core kernel, f16 = half-precision storage + v_mfma_f32_16x16x16_f16.\n"
            "# GB/s = ALGORITHMIC bytes P*(Cin*esz+4) + N*Cout*esz over the launch time; % of the 8 TB/s HBM3E peak.\n")
for src, dst, first in (
        ("conv_1m.txt", f"{tag}_conv_layers.txt",
         "# python tools/bench_conv.py                      (configs[1]: 1M-point tree, 2 cm voxels; 1 x MI355X)\n"),
        ("conv_5m.txt", f"{tag}_conv_layers_config4.txt",
         "# python tools/bench_conv.py 5000000 0.01 0.6      (configs[3]: 5M-point dense canopy, 1 cm voxels; 1 x MI355X)\n")):
    b = body(src)
    if b:
        (out / dst).write_text(first + CONV_HDR + b)
a, b = body("fp16_1m.txt"), body("fp16_5m.txt")
if a and b:
    (out / f"{tag}_config5_fp16.txt").write_text(
        "# python tools/time_fp16.py [5000000 0.01 0.6]    (configs[4]: peach-forest-65, float32 vs half-precision storage; "
        "1 x MI355X, one stream)\n# 1M-point tree, 2 cm voxels:\n" + a + "# 5M-point dense canopy, 1 cm voxels:\n" + b)
b = body(f"{tag}_free_running.txt")
if b:
    (out / f"{tag}_free_running.txt").write_text("# python bench.py --free-running --no-cpu-baseline --no-extras, twice on one box (tools/collect_r2.sh): the default schedule\n"
                                                 "# (chip-filling phases of the batches in flight take turns) beside the free-running one\n" + b)
solo = sorted(glob.glob(str(ROOT / "gpurun_out" / "prof_solo" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
if solo:
    shutil.copy(solo[-1], out / f"{tag}_kernel_stats_single_stream.csv")
print("wrote", sorted(p.name for p in out.iterdir()))
