"""world_size-2 test of the sharded MSM path on CPU: gloo for the all-gather, the test-only
emulation library for each rank's partial MSM.  On the GPU box the same host code runs with the
product library and backend nccl (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, cid, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mira_amd import _lib
    from mira_amd import commitment as cm
    from mira_amd.dist import ShardedCommitmentKey, chunk_bounds
    from oracle import cref as C
    lib = _lib.MiraLib(os.path.join(ROOT, "tests", "emu", "libmira_emu.so"))
    # all ranks must agree on the window width although their chunks of a prefix differ in length:
    # curve 0 pins it, curve 1 leaves it to the rule the sharded key derives from the global length
    key = ShardedCommitmentKey.synthetic(cid, n, lib=lib, window_bits=8 if cid == 0 else 0)
    lo, hi = chunk_bounds(n, world, rank)
    results = {}
    for n_commit in (n, n - 37, 5):                        # full key, a prefix cutting the last chunk, a prefix inside rank 0
        n_local = key.local_prefix(n_commit)
        d = cm.synth_scalars_device(cid, max(n_local, 1), index0=lo, lib=lib)
        results[n_commit] = key.commit_device(d, n_commit)
    try:
        key.commit_device(d, n + 1)
        results["too_long"] = False
    except cm.TooLongInput as e:
        results["too_long"] = (e.input_len, e.limit) == (n + 1, n)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack([results[n], results[n - 37], results[5]]))
    assert results["too_long"]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cid", [0, 1])
def test_sharded_commit_world2(tmp_path, emu_lib, cid):
    from oracle import cref as C
    world, n = 2, 301
    port = 29500 + (os.getpid() % 2000) + cid
    mp.spawn(_worker, args=(world, port, n, cid, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert (r0 == r1).all()                                # every rank holds the same commitment
    bases, sc = C.synth_bases(cid, n), C.synth_scalars(cid, n)
    for row, m in zip(r0, (n, n - 37, 5)):
        assert (row == C.commit(cid, bases, sc[:m])).all()


def test_chunk_bounds_cover_everything():
    from mira_amd.dist import chunk_bounds
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [chunk_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
