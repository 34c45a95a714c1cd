"""Copy the rocprofv3 outputs merged into gpurun_out/ into tracked, judged summaries under profiles/.

    python tools/summarize_profiles.py r01
"""
import collections, csv, glob, json, os, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)

bench = ROOT / "gpurun_out" / f"{tag}_bench.json"
if bench.exists():
    line = [l for l in bench.read_text().splitlines() if l.startswith("{")][-1]
    (out / f"{tag}_bench.json").write_text(json.dumps(json.loads(line), indent=1) + "\n")

stats = sorted(glob.glob(str(ROOT / "gpurun_out" / f"prof_{tag}" / "*" / "*kernel_stats.csv")), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], out / f"{tag}_kernel_stats.csv")


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
print("wrote", sorted(p.name for p in out.iterdir()))
