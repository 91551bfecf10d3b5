import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def free_port():
    """A TCP port nobody holds on 127.0.0.1 right now (rendezvous of the multi-process tests; a port derived from the pid
    collides between pytest-xdist workers)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host emulation of the kernel sources (tests/emu), for kernel-logic checks without
    a GPU.  Never used by the product package."""
    from mira_amd import _lib
    csrc = os.path.join(ROOT, "mira_amd", "csrc")
    import fcntl
    with open(os.path.join(csrc, ".emu_build.lock"), "w") as lock:      # pytest-xdist workers build one at a time
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-C", csrc, "emu"])
    return _lib.MiraLib(os.path.join(ROOT, "tests", "emu", "libmira_emu.so"))


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mira_amd import _lib
    return _lib.load()
