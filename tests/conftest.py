import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host emulation of the kernel sources (tests/emu), for kernel-logic checks without
    a GPU.  Never used by the product package."""
    from mira_amd import _lib
    csrc = os.path.join(ROOT, "mira_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "emu"])
    return _lib.MiraLib(os.path.join(ROOT, "tests", "emu", "libmira_emu.so"))


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mira_amd import _lib
    return _lib.load()
