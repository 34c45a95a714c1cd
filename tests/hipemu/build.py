"""TEST INFRASTRUCTURE: CPU sanitizer build of the HIP kernel sources (see include/hip/hip_runtime.h)."""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
CSRC = ROOT / "smart_tree_amd" / "csrc"
OUT = HERE / "_build"
LIB = OUT / "libsmarttree_emu.so"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unknown-attributes", "-Wno-unused-value"]


def build(force: bool = False) -> Path:
    OUT.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob("*.hip"))
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted((HERE / "include" / "hip").glob("*.h")) + \
        sorted((ROOT / "include").glob("*.h"))
    if not force and LIB.exists() and all(LIB.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return LIB
    newest_hdr = max(d.stat().st_mtime for d in deps if d.suffix == ".h")
    objs = []
    for src in srcs:
        obj = OUT / (src.stem + ".o")
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, newest_hdr):
            subprocess.run([CLANG, *FLAGS, "-I", str(HERE / "include"), "-I", str(ROOT / "include"),
                            "-I", str(CSRC), "-c", str(src), "-o", str(obj)], check=True)
        objs.append(str(obj))
    subprocess.run([CLANG, "-shared", "-fPIC", "-o", str(LIB), *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
