"""SURVEY.md 8(e) on the one-GPU box: the RCCL branch of the sharded commit with world size 1 in
a fresh child process, and the two-rank rehearsal of `bench.py --gpus 2` (both ranks on GPU 0, gloo exchange) --
the multi-rank host path with the product library, every point against the oracle.  The 8-GPU run itself is the
driver's; the world-size-2 gloo tests of tests/test_dist_gloo.py cover the exchange on CPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port
from oracle import cref as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("cid", [0, 1])
def test_sharded_commit_over_rccl_world1(gpu_lib, cid):
    n = 1 << 16
    port = str(free_port())
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_child.py"), str(n), str(cid), port], capture_output=True, text=True, timeout=600, env=clean_env())
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    bases, sc = C.synth_bases(cid, n), C.synth_scalars(cid, n)
    for m in (n, n - 1234, 7):
        assert (np.array(out["points"][str(m)], dtype=np.uint64) == C.msm_pippenger(cid, sc[:m], bases[:m])).all(), m
        agreed, used = out["widths"][str(m)]
        assert agreed == used                                   # the partial was cut with the width every rank derives
    assert out["too_long"] == [n + 1, n]
    assert out["single_gpu_planner_width"] == out["widths"][str(n)][0]   # ... which is the single-GPU planner's choice for that length


def test_bench_two_ranks_on_one_gpu(gpu_lib):
    """`python bench.py --gpus 2 --rehearse-one-gpu`: bench.py starts its own two ranks (before touching the GPU),
    each cuts its half of one 2^18 MSM on GPU 0, gloo carries the all-gather; the commitment every rank ends with
    is the oracle's."""
    log_n = 18
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--total-log-n", str(log_n), "--window-bits", "0",
                        "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu"], capture_output=True, text=True, timeout=900, env=clean_env())
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["scaling"] == "strong" and "REHEARSAL" in out["data"]
    n = 1 << log_n
    want = C.msm_pippenger(0, C.synth_scalars(0, n), C.synth_bases(0, n))
    assert [int(v, 16) for v in out["result_affine_u64"]] == [int(v) for v in want]
