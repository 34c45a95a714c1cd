import ctypes
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """CPU sanitizer build of the HIP kernel sources (tests/hipemu) -- test infrastructure only."""
    sys.path.insert(0, str(ROOT / "tests" / "hipemu"))
    import build as emu_build  # noqa

    from smart_tree_amd import _lib

    import os

    path = os.environ.get("SMARTTREE_EMU_LIB") or str(emu_build.build())  # sanitize.py points at its ASan/UBSan build
    return _lib.declare(ctypes.CDLL(path))


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    """Runs a kernel-level parity test twice: on the CPU sanitizer build (not gpu) and on the
    real HIP library on cuda:0 (gpu).  Yields the torch device tensors must live on."""
    from smart_tree_amd import _lib

    if request.param == "emu":
        monkeypatch.setattr(_lib, "_LIB", request.getfixturevalue("emu_lib"))
        monkeypatch.setattr(_lib, "_ALLOW_HOST_POINTERS", True)
        yield torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
        monkeypatch.setattr(_lib, "_LIB", None)
        _lib.lib()  # must load the real HIP extension; raises if missing
        yield torch.device("cuda:0")
