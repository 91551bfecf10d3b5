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
from harness import main_gate as MG
PLAN = MG.fold_step_msm_schedule(K)      # {curve: (witness length, cross terms)} derived from the reference's configure functions: (14 << K, 6), (7 << K, 5)
assert PLAN == {cm.CURVE_BN256: (14 << K, 6), cm.CURVE_GRUMPKIN: (7 << K, 5)}


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


def test_reference_largest_commit_14x2p24_and_k28_key_file(gpu_lib, tmp_path):
    """The reference's largest real sizes (examples/groth16/main.rs:47-75: k = 24 tables over keys of 2^27 .. 2^28 points):
    a witness commit of 14 x 2^24 = 234 881 024 pairs and a 2^28-point key through its cache file.
    * the commit under 13-bit windows has 4.7 G sorted entries -- more than the 32-bit offsets of one pass -- and is cut into
      point chunks inside the launch sequence; under 16-bit windows it fits one pass; eight chunk partials combined are the
      same point; two 2^22-pair chunk partials are their chunks' commitments by the oracle;
    * 2^28 pairs under 16-bit windows are exactly 2^32 entries: two passes; whole = two half partials combined;
    * the 2^28-point key written as the reference's raw `[C]` file (16 GiB) and read back by mira_msm_register_bases_file with
      the curve check: same commitment, same bytes at the chunk boundaries."""
    import time
    cid = 0
    n28, n = 1 << 28, 14 << 24
    key = cm.CommitmentKey.synthetic(cid, n28, seed=0x3238)
    d = cm.synth_scalars_device(cid, n28, seed=0x3239)
    timings = {}
    try:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(16))
        key.commit_device(d, 1 << 20)
        t0 = time.perf_counter(); whole = key.commit_device(d, n); timings["commit_14x2p24_c16_ms"] = (time.perf_counter() - t0) * 1e3
        G, per = 8, n // 8
        parts = []
        for g in range(G):
            part, c, w = key.commit_partial_device(g * per, d + g * per * 32, per)
            assert (c, w) == (16, 16)
            parts.append(part)
        assert (cm.combine_partials(cid, np.stack(parts), 16, 16) == whole).all()
        t0 = time.perf_counter(); whole28 = key.commit_device(d, n28); timings["commit_2p28_c16_ms"] = (time.perf_counter() - t0) * 1e3      # 2^32 entries: two passes
        halves = [key.commit_partial_device(h * (n28 // 2), d + h * (n28 // 2) * 32, n28 // 2)[0] for h in range(2)]
        assert (cm.combine_partials(cid, np.stack(halves), 16, 16) == whole28).all()
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(13))    # 20 windows: 4.7 G entries, two passes
        t0 = time.perf_counter(); w13 = key.commit_device(d, n); timings["commit_14x2p24_c13_ms"] = (time.perf_counter() - t0) * 1e3
        assert (w13 == whole).all()
        # the oracle on two chunks of 2^22 pairs: the first, and one that straddles the boundary of the two passes
        m = 1 << 22
        for first in (0, n // 2 - m // 2):
            part, c, w = key.commit_partial_device(first, d + first * 32, m, window_bits=13)
            want = C.commit(cid, key.download(first, m), gpu_lib.download(d + first * 32, (m, 4)))
            assert (cm.combine_partials(cid, part[None, :], c, w) == want).all()
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
    path = tmp_path / "28.bin"
    t0 = time.perf_counter(); key.save_to_file(path); timings["save_2p28_s"] = time.perf_counter() - t0
    assert path.stat().st_size == n28 * 64
    key.close()
    t0 = time.perf_counter(); loaded = cm.CommitmentKey.load_from_file(cid, path, 28, validate=True); timings["load_2p28_validated_s"] = time.perf_counter() - t0
    try:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(16))
        assert (loaded.commit_device(d, n) == whole).all()
        ref = cm.CommitmentKey.synthetic(cid, 4096, seed=0x3238, index0=(1 << 27) - 2048)
        assert (loaded.download((1 << 27) - 2048, 4096) == ref.download()).all()
        ref.close()
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
        loaded.close(); gpu_lib.free(d)
        path.unlink()
    gpu_lib.check(gpu_lib.c.mira_trim(0, None))
    print("largest-size timings:", {k: round(v, 3) for k, v in timings.items()})
