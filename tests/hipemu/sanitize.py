"""TEST INFRASTRUCTURE: run a subset of the kernel-parity tests with the CPU sanitizer build compiled with
AddressSanitizer + UndefinedBehaviorSanitizer (the GPU pool offers neither).

    python tests/hipemu/sanitize.py            # builds tests/hipemu/_build_asan/libsmarttree_emu.so, runs pytest under it
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
import build as emu_build  # noqa: E402

CLANG = emu_build.CLANG


def main():
    out = HERE / "_build_asan"
    out.mkdir(exist_ok=True)
    flags = [f for f in emu_build.FLAGS if f not in ("-O1",)] + ["-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                                                                 "-fno-sanitize-recover=undefined"]
    objs = []
    for src in sorted(emu_build.CSRC.glob("*.hip")):
        obj = out / (src.stem + ".o")
        subprocess.run([CLANG, *flags, "-I", str(HERE / "include"), "-I", str(ROOT / "include"), "-I", str(emu_build.CSRC),
                        "-c", str(src), "-o", str(obj)], check=True)
        objs.append(str(obj))
    lib = out / "libsmarttree_emu.so"
    subprocess.run([CLANG, "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan", "-o", str(lib), *objs], check=True)
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1", SMARTTREE_EMU_LIB=str(lib))
    tests = sys.argv[1:] or ["tests/test_prims.py", "tests/test_voxelize.py", "tests/test_skeleton.py", "tests/test_golden.py",
                             "tests/test_pipeline.py"]
    return subprocess.run([sys.executable, "-m", "pytest", *tests, "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"],
                          cwd=ROOT, env=env).returncode


if __name__ == "__main__":
    sys.exit(main())
