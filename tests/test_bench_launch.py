"""`python bench.py --gpus N` must itself start N ranks (the driver's multi-GPU command carries no
launcher).  Rehearsed here on CPU: gloo + the test-only host emulation of the kernels (--emulate),
tiny sizes.  On the GPU box the same code path runs with RCCL and the product library."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                      # the contract: ONE JSON line on stdout
    return json.loads(lines[0])


def test_bench_gpus2_launches_two_ranks(emu_lib):
    out = run_bench("--gpus", "2", "--emulate", "--total-log-n", "8", "--window-bits", "8", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu")
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["scaling"] == "strong" and out["config"]["total_pairs"] == 256 and out["config"]["pairs_per_gpu"] == 128
    assert out["ms_per_step"] > 0 and "EMULATION" in out["data"]
    # the line diagnoses itself: what every rank held and measured, and that all ranks combined partials of one shape
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["pairs"] == 128 for r in out["ranks"])
    assert all(r["partial_ms"] > 0 and r["exchange_us"] > 0 and r["combine_ms"] > 0 for r in out["ranks"])
    assert out["agreed_window"] == {"window_bits": 8, "num_windows": 32, "same_on_all_ranks": True}


def test_bench_gpus2_weak_variant(emu_lib):
    out = run_bench("--gpus", "2", "--emulate", "--log-n", "7", "--window-bits", "8", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["total_pairs"] == 256
