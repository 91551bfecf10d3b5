"""Point-chunk sharding of one MSM across the GPUs of a node: one process per GPU
(`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The (scalar, base) pairs are cut into `world` contiguous chunks; rank g keeps the bases of chunk g
resident on its GPU and reduces its chunk to W window sums.  The only exchange is one all-gather of
W * 128 bytes per rank (elliptic-curve addition is not an RCCL reduction operator, so
all-gather + local add is the "reduce"); every rank then holds the same affine commitment.
Message size is <= 8 KiB per rank: latency-bound, xGMI bandwidth is irrelevant.
"""
import ctypes

import numpy as np

from . import _lib
from .commitment import CommitmentKey, TooLongInput, combine_partials


def chunk_bounds(n, world, rank):
    """Contiguous chunk [lo, hi) of rank `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedCommitmentKey:
    """`CommitmentKey` whose bases are spread over the ranks of a process group.

    `local_key` holds this rank's chunk [lo, hi) of a key of global length `total_len`.
    `commit_device(d_scalars_local, n_global)` commits the first n_global scalars of the global
    vector; this rank passes the device pointer of ITS part of that prefix."""

    def __init__(self, curve, local_key, total_len, group=None, lib=None, window_bits=0):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.curve = curve
        self.key = local_key
        self.total_len = total_len
        self.lo, self.hi = chunk_bounds(total_len, self.world, self.rank)
        self.lib = lib or local_key.lib
        self.window_bits = window_bits        # 0: derived from the global length (see _agreed_window_bits)
        self.check_shapes = True              # one 16-byte all-gather per commit: every rank's partial has the same (width, windows)
        self.last = {}                        # diagnostics of the most recent commit_device on THIS rank (bench.py gathers them)
        assert len(local_key) == self.hi - self.lo, "local key does not match this rank's chunk"

    @classmethod
    def synthetic(cls, curve, total_len, group=None, seed=0x42415345, lib=None, window_bits=0):
        import torch.distributed as dist
        lo, hi = chunk_bounds(total_len, dist.get_world_size(group), dist.get_rank(group))
        return cls(curve, CommitmentKey.synthetic(curve, hi - lo, seed=seed, index0=lo, lib=lib), total_len, group, lib, window_bits)

    def len(self):
        return self.total_len

    def local_prefix(self, n_global):
        """How many of the first n_global pairs live on this rank."""
        return max(0, min(self.hi, n_global) - self.lo)

    def _agreed_window_bits(self, n_global):
        """Every rank must cut its scalars into the same windows, but the ranks' chunk lengths differ
        (a prefix of the key ends inside one rank's chunk), so the width cannot be left to each
        rank's planner with its own length and statistics: it is the planner's choice for the LARGEST local
        prefix any rank holds -- rank 0's, min(n_global, its chunk length): the key is cut by its TOTAL length, so a
        short prefix lies in the first chunks only -- which all ranks compute alike from (n_global, total_len, world)
        (mira_msm_plan_window_bits is a pure function of the length)."""
        if self.window_bits:
            return self.window_bits
        lo0, hi0 = chunk_bounds(self.total_len, self.world, 0)
        c = ctypes.c_int32()
        self.lib.check(self.lib.c.mira_msm_plan_window_bits(max(1, min(n_global, hi0 - lo0)), ctypes.byref(c)))
        return c.value

    def _assert_same_shape(self, c, w, n_local, device):
        """Partials of different shapes cannot be combined, and a data all-gather of different lengths would hang or
        corrupt: before the exchange every rank publishes (width, windows) and all must agree (they do by construction;
        a rank whose key has tables the others lack, or a different forced width, fails here with the ranks named)."""
        import torch
        mine = torch.tensor([c, w], dtype=torch.int64, device=device)
        out = torch.empty(2 * self.world, dtype=torch.int64, device=device)
        self.dist.all_gather_into_tensor(out, mine, group=self.group)
        shapes = [tuple(int(v) for v in out[2 * r: 2 * r + 2].tolist()) for r in range(self.world)]
        if len(set(shapes)) != 1:
            raise RuntimeError(f"ranks disagree on the shape of their partials (window bits, windows) by rank: {shapes}; rank {self.rank} holds {n_local} pairs")

    def commit_device(self, d_scalars_local, n_global):
        """src/commitment.rs:78-87 over the sharded key; every rank returns the same point.
        Fixed-base tables (`precompute()`) must be built on all ranks or on none.

        The partial stays in device memory: mira_msm_partial_to_device writes the window sums into the tensor the
        all-gather reads (RCCL over xGMI), and ONE device-to-host copy of world * W * 128 bytes follows the gather;
        the G - 1 additions per window, Horner and to_affine run on the host (mira_msm_combine)."""
        import time
        import torch
        if n_global > self.total_len:
            raise TooLongInput(n_global, self.total_len)
        n_local = self.local_prefix(n_global)
        t_start = time.perf_counter()
        # table partials have one layout whatever the length; otherwise every rank names the same width
        tables = getattr(self.key, "precomputed", False)
        width = 0 if tables else self._agreed_window_bits(n_global)
        words = _lib.MIRA_PARTIAL_U64
        if self.dist.get_backend(self.group) == "nccl":
            mine = torch.empty(words, dtype=torch.int64, device="cuda")
            c, w = self.key.commit_partial_to_device(0, d_scalars_local, n_local, mine.data_ptr(), window_bits=width)   # synchronises the library's stream
            t_partial = time.perf_counter()
            if self.check_shapes:
                self._assert_same_shape(c, w, n_local, "cuda")
            out = torch.empty(self.world * w * 16, dtype=torch.int64, device="cuda")
            self.dist.all_gather_into_tensor(out, mine[: w * 16], group=self.group)
            gathered = out.cpu().numpy().view(np.uint64).reshape(self.world, w * 16)
        else:                                                  # gloo (CPU tests, one-GPU rehearsal): the same partial, through host memory
            d_part = self.lib.alloc(words * 8)
            try:
                c, w = self.key.commit_partial_to_device(0, d_scalars_local, n_local, d_part, window_bits=width)
                mine = torch.from_numpy(self.lib.download(d_part, (words,))[: w * 16].view(np.int64).copy())
            finally:
                self.lib.free(d_part)
            t_partial = time.perf_counter()
            if self.check_shapes:
                self._assert_same_shape(c, w, n_local, "cpu")
            out = torch.empty(self.world * w * 16, dtype=torch.int64)
            self.dist.all_gather_into_tensor(out, mine, group=self.group)
            gathered = out.numpy().view(np.uint64).reshape(self.world, w * 16)
        t_exchanged = time.perf_counter()
        parts = np.zeros((self.world, words), dtype=np.uint64)
        parts[:, : w * 16] = gathered
        point = combine_partials(self.curve, parts, c, w, lib=self.lib)
        t_done = time.perf_counter()
        self.last = {"rank": self.rank, "pairs": int(n_local), "window_bits": int(c), "num_windows": int(w),
                     "partial_ms": round((t_partial - t_start) * 1e3, 4),          # this rank's chunk: kernels + the synchronisation that ends them
                     "exchange_us": round((t_exchanged - t_partial) * 1e6, 1),      # shape check + all-gather + the copy of world * W * 128 bytes to the host (includes waiting for the slowest rank)
                     "combine_ms": round((t_done - t_exchanged) * 1e3, 4)}          # (world - 1) * W additions, Horner, to_affine on the host
        return point
