"""Compile the HIP sources into smart_tree_amd/libsmarttree_hip.so for gfx950.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box
with the gpurun snapshot.  Rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libsmarttree_hip.so"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(CSRC.glob("*.hip"))


def build(force: bool = False, verbose: bool = False) -> Path:
    srcs = sources()
    deps = srcs + sorted(CSRC.glob("*.h")) + sorted((PKG.parent / "include").glob("*.h"))
    if not force and LIB.exists() and all(LIB.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return LIB
    objs = []
    for src in srcs:
        obj = CSRC / (src.stem + ".o")
        hdr_new = any(h.stat().st_mtime > (obj.stat().st_mtime if obj.exists() else 0) for h in deps if h.suffix == ".h")
        if force or hdr_new or not obj.exists() or obj.stat().st_mtime < src.stat().st_mtime:
            cmd = ["hipcc", *FLAGS, "-I", str(PKG.parent / "include"), "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(str(obj))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
