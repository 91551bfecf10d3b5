"""GPU parity of the two BASELINE configurations round 1 only covered from bench.py / tools:

* configs[3] -- the MSM schedule of one IVC fold step at k = 17: per curve one
  witness commit of 14 * 2^17 (BN256) / 7 * 2^17 (Grumpkin) witness-like scalars
  (src/plonk/mod.rs:680-688) and the 6 / 5 cross-term commits of 2^17 uniform scalars that
  `commit_cross_terms` issues one by one (src/nifs/vanilla/mod.rs:123-127).  Every one of the 13
  points is checked against the oracle, and the batched submission against one call per commit.
* configs[4] on one GPU -- a 2^26 MSM (src/commitment.rs:78-87): the whole commit, its eight
  point-chunk partials combined, and the oracle's `best_multiexp` agree bit for bit.
"""
import numpy as np
import pytest

from mira_amd import commitment as cm
from oracle import cref as C

pytestmark = pytest.mark.gpu

K = 17
PLAN = {cm.CURVE_BN256: (14 << K, 6), cm.CURVE_GRUMPKIN: (7 << K, 5)}   # (witness length, cross terms)


def test_fold_step_k17_schedule(gpu_lib):
    n = 1 << K
    calls = 0
    for cid, (nw, cnt) in PLAN.items():
        key = cm.CommitmentKey.synthetic(cid, nw, seed=0x464F4C44 + cid)
        bases = key.download()
        d_wit = cm.synth_scalars_device(cid, nw, seed=0x1000 + cid, kind=1)
        wit = gpu_lib.download(d_wit, (nw, 4))
        cross = [C.synth_scalars(cid, n, seed=0x2000 + 16 * cid + i) for i in range(cnt)]
        # the reference's order: witness commit first, then the cross terms one after another
        got_w = key.commit(wit)                                  # host scalars, as commit(&self, v: &[C::Scalar]) receives them
        seq = [key.commit(v) for v in cross]
        calls += 1 + cnt
        assert (got_w == C.commit(cid, bases, wit)).all(), f"witness commit, curve {cid}"
        assert (key.commit_device(d_wit, nw) == got_w).all()
        for i, v in enumerate(cross):
            assert (seq[i] == C.commit(cid, bases[:n], v)).all(), f"cross term {i}, curve {cid}"
        bat = key.commit_batch(cross)                            # one submission for all cross terms
        assert (bat == np.stack(seq)).all()
        d_cross = gpu_lib.alloc(cnt * n * 32)
        for i, v in enumerate(cross):
            gpu_lib.upload(d_cross + i * n * 32, v)
        assert (key.commit_batch_device(d_cross, n, cnt) == bat).all()
        gpu_lib.free(d_cross); gpu_lib.free(d_wit); key.close()
    assert calls == 13


def test_msm_2p26_whole_partials_oracle(gpu_lib):
    cid, log_n, G = 0, 26, 8
    n = 1 << log_n
    key = cm.CommitmentKey.synthetic(cid, n, seed=0x3236)
    d = cm.synth_scalars_device(cid, n, seed=0x3237)
    gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(16))
    try:
        whole = key.commit_device(d, n)
        per = n // G
        parts = []
        for g in range(G):                                       # what the 8 ranks of configs[4] compute
            part, c, w = key.commit_partial_device(g * per, d + g * per * 32, per)
            assert (c, w) == (16, 16)
            parts.append(part)
        assert (cm.combine_partials(cid, np.stack(parts), c, w) == whole).all()
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(13))   # another width, the same group element
        assert (key.commit_device(d, n) == whole).all()
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
    # oracle: per-chunk MSMs summed on the host (each chunk is an independent best_multiexp call,
    # so the host never holds more than one chunk of the 6 GiB of inputs)
    acc = np.zeros(8, dtype=np.uint64)
    for g in range(G):
        bases = key.download(g * per, per)
        sc = gpu_lib.download(d + g * per * 32, (per, 4))
        acc = C.ec_add(cid, acc, C.commit(cid, bases, sc))
        # and the GPU's own partial of this chunk is that chunk's commitment
        assert (cm.combine_partials(cid, parts[g][None, :], 16, 16) == C.commit(cid, bases, sc)).all()
    assert (acc == whole).all()
    gpu_lib.free(d); key.close()
