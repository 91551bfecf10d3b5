"""The compiled-C consumer of include/mira_gpu.h against the product library on the GPU:
tests/abi/consumer.c, built here by the system C compiler and linked to mira_amd/csrc/libmira_gpu.so, run as a child
process: register -> mira_msm on (r - 1) * G -> unregister, TooLongInput, the 8-point FFT vector, a compiled graph,
the key-file error path, mira_trim."""
import subprocess

import pytest

from test_abi import build_c_consumer

pytestmark = pytest.mark.gpu


def test_c_consumer_against_libmira_gpu(gpu_lib, tmp_path):
    exe = build_c_consumer(gpu_lib.path, tmp_path / "consumer_gpu")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "consumer ok" in res.stdout, res.stdout + res.stderr
