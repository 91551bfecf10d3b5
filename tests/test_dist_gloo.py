"""world_size-2 test of the sharded MSM path on CPU: gloo for the all-gather, the test-only
emulation library for each rank's partial MSM.  On the GPU box the same host code runs with the
product library and backend nccl (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import free_port

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
    port = free_port()
    mp.spawn(_worker, args=(world, port, n, cid, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert (r0 == r1).all()                                # every rank holds the same commitment
    bases, sc = C.synth_bases(cid, n), C.synth_scalars(cid, n)
    for row, m in zip(r0, (n, n - 37, 5)):
        assert (row == C.commit(cid, bases, sc[:m])).all()


def _worker8(rank, world, port, n, out_dir):
    """Eight ranks, uneven chunks: the full key, a prefix that ends inside rank 0's chunk, one that ends inside rank 5's;
    then the same with a shared-bucket table set on every rank; then a table set on the odd ranks only -- every rank must
    refuse in the shape check (nobody hangs in a mismatched all-gather)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mira_amd import _lib
    from mira_amd import commitment as cm
    from mira_amd.dist import ShardedCommitmentKey, chunk_bounds
    lib = _lib.MiraLib(os.path.join(ROOT, "tests", "emu", "libmira_emu.so"))
    lib.tune(_lib.TUNE_SHARED_MIN_N, 1)
    cid = 0
    lo, hi = chunk_bounds(n, world, rank)
    lo0, hi0 = chunk_bounds(n, world, 0)
    lo5, hi5 = chunk_bounds(n, world, 5)
    commits = (n, hi0 - 3, lo5 + 2)
    rows, diag = [], []
    key = ShardedCommitmentKey.synthetic(cid, n, lib=lib, window_bits=7)
    d = cm.synth_scalars_device(cid, hi - lo, index0=lo, lib=lib)
    for m in commits:
        rows.append(key.commit_device(d, m))
        diag.append(dict(key.last))
        assert key.last["pairs"] == key.local_prefix(m) and (key.last["window_bits"], key.last["num_windows"]) == (7, 37)
        assert all(k in key.last for k in ("partial_ms", "exchange_us", "combine_ms"))
    # the planner's width for the largest local prefix (rank 0's), alike on all ranks although their own lengths differ
    auto = ShardedCommitmentKey(cid, key.key, n, lib=lib, window_bits=0)
    widths = {auto._agreed_window_bits(m) for m in commits}
    import ctypes
    for m in commits:
        c = ctypes.c_int32()
        lib.check(lib.c.mira_msm_plan_window_bits(max(1, min(m, hi0 - lo0)), ctypes.byref(c)))
        assert auto._agreed_window_bits(m) == c.value
    # tables on every rank: table partials (width 0, one sum), same points
    key.key.precompute(9)
    for m in commits:
        rows.append(key.commit_device(d, m))
        assert (key.last["window_bits"], key.last["num_windows"]) == (0, 1)
    # tables on the odd ranks only: the shapes differ, every rank raises before the data exchange
    mixed = ShardedCommitmentKey(cid, key.key if rank % 2 else cm.CommitmentKey.synthetic(cid, hi - lo, index0=lo, lib=lib), n, lib=lib, window_bits=0)
    try:
        mixed.commit_device(d, n)
        refused = False
    except RuntimeError as e:
        refused = "disagree" in str(e)
    assert refused
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack(rows))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_commit_world8(tmp_path, emu_lib):
    """The strong-scaling path of bench.py --gpus 8 on eight CPU ranks (gloo + the emulation library)."""
    from oracle import cref as C
    world, n, cid = 8, 203, 0
    port = free_port()
    mp.spawn(_worker8, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"rank{r}.npy") for r in range(world)]
    assert all((g == got[0]).all() for g in got)          # every rank holds the same commitments
    from mira_amd.dist import chunk_bounds
    hi0, lo5 = chunk_bounds(n, world, 0)[1], chunk_bounds(n, world, 5)[0]
    bases, sc = C.synth_bases(cid, n), C.synth_scalars(cid, n)
    want = [C.commit(cid, bases, sc[:m]) for m in (n, hi0 - 3, lo5 + 2)]
    for k, row in enumerate(got[0]):
        assert (row == want[k % 3]).all(), k


def test_chunk_bounds_cover_everything():
    from mira_amd.dist import chunk_bounds
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [chunk_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
