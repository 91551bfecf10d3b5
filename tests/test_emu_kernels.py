"""Kernel-logic checks without a GPU: the same kernel sources built against the test-only host
emulation (tests/emu/emu.h), compared with the oracle.  Sizes are tiny: every emulated lane is
a host loop iteration or an OS thread.  The GPU parity tests proper are test_gpu_*.py."""
import ctypes

import numpy as np
import pytest

from helpers import golden_msm_case, ints_to_mont, load_golden, mont_to_ints
from mira_amd import commitment as cm
from mira_amd import fft as F
from mira_amd import _lib
from oracle import cref as C
from oracle import pyref as P



@pytest.fixture
def tune(emu_lib):
    """mira_set_tuning for the duration of one test."""
    used = []

    def set_knob(knob, value):
        used.append(knob)
        emu_lib.tune(knob, value)
    yield set_knob
    for knob in used:
        emu_lib.tune(knob, -1)


def test_emu_synth_matches_oracle(emu_lib):
    for cid in (0, 1):
        p = cm.synth_scalars_device(cid, 70, seed=3, kind=1, lib=emu_lib)
        assert (emu_lib.download(p, (70, 4)) == C.synth_scalars(cid, 70, seed=3, kind=1)).all()
        key = cm.CommitmentKey.synthetic(cid, 6, seed=4, lib=emu_lib)
        assert (key.bases() == C.synth_bases(cid, 6, seed=4)).all()
        key.check_on_curve()


def test_emu_check_on_curve_rejects_bad_point(emu_lib):
    from mira_amd._lib import MiraError, MIRA_E_INVALID_POINT
    bs = C.synth_bases(0, 4)
    bs[2, 0] ^= np.uint64(1)
    key = cm.CommitmentKey(0, bs, lib=emu_lib)
    with pytest.raises(MiraError) as e:
        key.check_on_curve()
    assert e.value.code == MIRA_E_INVALID_POINT


@pytest.mark.parametrize("cid", [0, 1])
def test_emu_msm_golden(emu_lib, cid):
    for case in load_golden("msm_vectors.json")[str(cid)]:
        if case["n"] > 33:
            continue
        sc, bs, expected = golden_msm_case(cid, case)
        key = cm.CommitmentKey(cid, bs, lib=emu_lib)
        assert (key.commit(sc) == expected).all()


@pytest.mark.parametrize("c", [6])
def test_emu_msm_skewed_and_forced_window(emu_lib, c):
    """Small windows make every bucket heavy: exercises cut runs, the short fix-up chain and the
    workgroup-wide heavy fix-up."""
    cid, n = 0, 900
    bs = C.synth_bases(cid, 16)
    bs = np.tile(bs, (n // 16 + 1, 1))[:n]
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
    try:
        ones = np.tile(C.to_mont(C.FIELD_FR, np.array([1, 0, 0, 0], dtype=np.uint64)), (n, 1))
        for sc in (ones, C.synth_scalars(cid, n, seed=8, kind=1), C.synth_scalars(cid, n, seed=9)):
            assert (key.commit(sc) == C.msm_pippenger(cid, sc, bs)).all()
    finally:
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))


@pytest.mark.parametrize("cid", [0, 1])
def test_emu_bucket_reduction_shapes(emu_lib, tune, cid):
    """reduce_kernels.cuh: the weighted bucket sum as running sums over chunks + a tree whose nodes carry the per-bit sums,
    delivered in P pieces to the host's chain of doublings.  Every shape gives the oracle's point: pieces 1 .. 4, chunks of
    2 / 4 / 8 buckets, phase A by quads and by single lanes, widths whose trees have zero, one and several levels in the
    second kernel; single commits, a batch and a shared-bucket table set."""
    n = 220
    bs, sc = C.synth_bases(cid, n, seed=61), C.synth_scalars(cid, n, seed=62)
    sc[3] = 0
    sc[5] = sc[6]
    bs[9] = bs[8]
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    want = C.msm_pippenger(cid, sc, bs)
    d = emu_lib.alloc(2 * n * 32)
    emu_lib.upload(d, sc); emu_lib.upload(d + n * 32, sc[::-1].copy())
    want_b = np.stack([want, C.msm_pippenger(cid, sc[::-1].copy(), bs)])
    try:
        for c, shapes in ((4, [(1, 1, 1), (2, 2, 0)]), (7, [(1, 2, 1), (3, 1, 0), (2, 3, 1)]), (10, [(4, 2, 1), (1, 3, 0), (3, 2, 0)])):
            emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
            for pieces, lam, quad in shapes:
                tune(_lib.TUNE_REDUCE_PIECES, pieces); tune(_lib.TUNE_REDUCE_LAMBDA, lam); tune(_lib.TUNE_REDUCE_QUAD, quad)
                assert (key.commit_device(d, n) == want).all(), (c, pieces, lam, quad)
                assert (key.commit(sc) == want).all()
            assert (key.commit_batch_device(d, n, 2) == want_b).all()
            part, cc, ww = key.commit_partial_device(0, d, n)                  # the public partial format: one point per window
            assert (cc, ww) == (c, -(-256 // c))
            assert (cm.combine_partials(cid, part[None, :], cc, ww, lib=emu_lib) == want).all()
    finally:
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))
    tune(_lib.TUNE_SHARED_MIN_N, 1)
    key.precompute(9)
    for pieces, lam, quad in ((1, 2, 1), (3, 3, 0), (4, 1, 1)):
        tune(_lib.TUNE_REDUCE_PIECES, pieces); tune(_lib.TUNE_REDUCE_LAMBDA, lam); tune(_lib.TUNE_REDUCE_QUAD, quad)
        assert (key.commit_device(d, n) == want).all(), (pieces, lam, quad)
    assert (key.commit_batch_device(d, n, 2) == want_b).all()
    emu_lib.free(d)


def test_emu_partial_and_combine(emu_lib):
    cid, n = 1, 600
    bs, sc = C.synth_bases(cid, n, seed=2), C.synth_scalars(cid, n, seed=3)
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    d = emu_lib.alloc(n * 32)
    emu_lib.upload(d, sc)
    want = C.msm_pippenger(cid, sc, bs)
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(9))
    parts, widths = [], set()
    for first, cnt in ((0, 250), (250, 350)):
        part, c, w = key.commit_partial_device(first, d + first * 32, cnt)
        parts.append(part); widths.add((c, w))
    assert widths == {(9, 29)}
    assert (cm.combine_partials(cid, np.stack(parts), c, w, lib=emu_lib) == want).all()
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))
    # without a forced width a partial is cut with 16 bits whatever its chunk length (an empty chunk
    # reports the plan without running it: 2^15 emulated buckets per window are slow; the GPU suite
    # combines unforced partials)
    assert {key.commit_partial_device(first, d, 0)[1:] for first in (0, 250)} == {(16, 16)}


@pytest.mark.parametrize("k", [0, 1, 3, 4, 10, 12])
def test_emu_ntt(emu_lib, k):
    a = C.synth_scalars(0, 1 << k, seed=1000 + k)
    assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all()
    assert (F.ifft(a, k, lib=emu_lib) == C.ifft(a, k)).all()
    if k in (4, 10):
        g = load_golden("ntt_vectors.json")[str(k)]
        assert mont_to_ints(F.coset_fft(a, lib=emu_lib), P.R_MOD) == [int(v, 16) for v in g["coset_fft"]]
        assert mont_to_ints(F.coset_ifft(a, lib=emu_lib), P.R_MOD) == [int(v, 16) for v in g["coset_ifft"]]


def test_emu_ntt_single_line_of_4096_points(emu_lib, tune):
    """2^10 .. 2^12 points take two passes since round 4 (one workgroup walking twelve layers was 2.3 x slower than 64 of them walking
    six twice); MIRA_TUNE_NTT_MAX_LOG_LINE = 12 still reaches the single 4096-point line (128 KiB of LDS, twiddles of nine layers in LDS)."""
    tune(_lib.TUNE_NTT_MAX_LOG_LINE, 12)
    for k in (12, 11):
        a = C.synth_scalars(0, 1 << k, seed=1100 + k)
        assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all()
        assert (F.ifft(a, k, lib=emu_lib) == C.ifft(a, k)).all()


def test_emu_ntt_four_step(emu_lib):
    k = 13   # two passes (column / twiddle / row); 2^10 .. 2^12 take the same route with shorter lines
    a = C.synth_scalars(0, 1 << k, seed=5)
    assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all()
    w = C.get_omega_or_inv(k, True)
    assert (F.best_fft(a, w, k, lib=emu_lib) == C.best_fft(a, w, k)).all()


@pytest.mark.parametrize("wave", [1, 0])
@pytest.mark.parametrize("max_line,ks", [(3, (4, 5, 6, 7, 8, 9)), (4, (9, 11, 12)), (2, (5, 6))])
def test_emu_ntt_pass_schedules(emu_lib, max_line, ks, wave):
    """Two- and three-pass schedules at emulation sizes: MIRA_TUNE_NTT_MAX_LOG_LINE shortens the
    lines so that every split is exercised, on the wave-level kernel (k_ntt_wave: 2^13 .. 2^24 on the
    GPU) and on the workgroup-level one (k_ntt_lines: 2^25 .. 2^28)."""
    emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, max_line)
    emu_lib.tune(_lib.TUNE_NTT_WAVE, wave)
    try:
        for k in ks:
            a = C.synth_scalars(0, 1 << k, seed=2000 + k)
            assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all(), k
            assert (F.ifft(a, k, lib=emu_lib) == C.ifft(a, k)).all(), k
        w = C.get_omega_or_inv(ks[-1], True)
        a = C.synth_scalars(0, 1 << ks[-1], seed=7)
        assert (F.best_fft(a, w, ks[-1], lib=emu_lib) == C.best_fft(a, w, ks[-1])).all()
        with pytest.raises(_lib.MiraError):
            F.fft(C.synth_scalars(0, 1 << (3 * max_line + 1), seed=1), 3 * max_line + 1, lib=emu_lib)
    finally:
        emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1)
        emu_lib.tune(_lib.TUNE_NTT_WAVE, -1)


@pytest.mark.parametrize("wave", [1, 0])
@pytest.mark.parametrize("grid", [3, 8, 13])
def test_emu_ntt_block_groups_from_counters(emu_lib, grid, wave):
    """Both NTT kernels hand the block-groups (k_ntt_wave) / lines (k_ntt_lines) of a pass out through per-XCD counters once a
    workgroup has four of them or more (ntt_kernels.cuh: nttw_first, nttw_grab).  On the GPU that starts at 2^22 and 2^25 points;
    MIRA_TUNE_NTT_GRID makes the grid small enough for the emulation: 3 workgroups (five of the eight ranges have no home
    workgroup: their block-groups are all taken by workgroups that walk on from their own range), 8 (one per range) and 13 (uneven
    homes) -- 2^15 points in three passes of 32-point lines are 32 block-groups (1 024 lines) per pass, 2^13 are eight block-groups
    (the static stride of the wave-level kernel)."""
    # the workgroup-level kernel is a workgroup per line on the emulation: 16-point lines, 256 of them per pass of 2^12 points
    max_line, sizes, inv = (5, (15, 13), 14) if wave else (4, (12,), 11)
    emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, max_line)
    emu_lib.tune(_lib.TUNE_NTT_WAVE, wave)
    emu_lib.tune(_lib.TUNE_NTT_GRID, grid)
    try:
        for k in sizes:
            a = C.synth_scalars(0, 1 << k, seed=4100 + k + grid)
            assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all(), k
        a = C.synth_scalars(0, 1 << inv, seed=4200 + grid)
        assert (F.ifft(a, inv, lib=emu_lib) == C.ifft(a, inv)).all()
    finally:
        emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1)
        emu_lib.tune(_lib.TUNE_NTT_WAVE, -1)
        emu_lib.tune(_lib.TUNE_NTT_GRID, -1)


@pytest.mark.parametrize("wave", [1, 0])
def test_emu_ntt_full_twiddle_table(emu_lib, wave):
    """The first post-twiddle from a table of n entries (on the GPU: 2^17 .. 2^24 points; here
    MIRA_TUNE_NTT_SINGLE_TW_LOG = 3 makes every split transform take it), beside a one-table and a
    two-table second boundary; forward and inverse transforms of three sizes alternate through the
    four cached table sets (one is evicted and rebuilt)."""
    emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, 3)
    emu_lib.tune(_lib.TUNE_NTT_WAVE, wave)
    emu_lib.tune(_lib.TUNE_NTT_SINGLE_TW_LOG, 3)
    try:
        for rep in range(2):
            for k in (5, 8, 9):                              # two passes; three passes with a one-table / two-table second boundary
                a = C.synth_scalars(0, 1 << k, seed=2500 + k + rep)
                assert (F.fft(a, k, lib=emu_lib) == C.fft(a, k)).all(), k
                assert (F.ifft(a, k, lib=emu_lib) == C.ifft(a, k)).all(), k
    finally:
        emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1)
        emu_lib.tune(_lib.TUNE_NTT_WAVE, -1)
        emu_lib.tune(_lib.TUNE_NTT_SINGLE_TW_LOG, -1)


@pytest.mark.parametrize("k", [6, 7, 8])
def test_emu_ntt_wave_full_lines(emu_lib, k):
    """k_ntt_wave with lines of 64, 128 and 256 points (one to three register/LDS transposes), as a
    single pass and as the first pass of a split."""
    for kk in (k, k + 3):
        emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, k)
        try:
            a = C.synth_scalars(0, 1 << kk, seed=3000 + kk)
            assert (F.fft(a, kk, lib=emu_lib) == C.fft(a, kk)).all(), kk
            assert (F.ifft(a, kk, lib=emu_lib) == C.ifft(a, kk)).all(), kk
        finally:
            emu_lib.tune(_lib.TUNE_NTT_MAX_LOG_LINE, -1)


def test_emu_omega(emu_lib):
    for k in (0, 3, 24, 28):
        for inv in (False, True):
            assert (F.get_omega_or_inv(k, inv, lib=emu_lib) == C.get_omega_or_inv(k, inv)).all()


def test_emu_commit_batch(emu_lib):
    """Batched cross-term commits (src/nifs/vanilla/mod.rs:124-127): each result equals its own commit."""
    for cid in (0, 1):
        n = 150
        bs = C.synth_bases(cid, n + 20, seed=6)
        key = cm.CommitmentKey(cid, bs, lib=emu_lib)
        vs = [C.synth_scalars(cid, n, seed=20 + i, kind=i % 2) for i in range(3)]
        got = key.commit_batch(vs)
        for i, v in enumerate(vs):
            assert (got[i] == C.commit(cid, bs, v)).all()
        assert (got[0] == key.commit(vs[0])).all()
        assert key.commit_batch([]).shape == (0, 8)
        with pytest.raises(cm.TooLongInput):
            key.commit_batch([C.synth_scalars(cid, n + 21)])


def test_emu_fixed_base_tables(emu_lib, tune):
    """mira_msm_precompute: window tables 2^(20 w) P_i, one shared set of 2^19 buckets, staged
    sort.  Same points as the per-window path and the oracle, with an identity base, a heavy
    bucket and chunk partials.  (One curve only: 2^19 emulated buckets are slow; the GPU suite
    covers both.)"""
    tune(_lib.TUNE_TABLE_MIN_N, 1)
    cid, n = 1, 200
    bs = C.synth_bases(cid, n, seed=40)
    bs[97] = 0
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    sc = C.synth_scalars(cid, n, seed=41, kind=1)
    sc[100:140] = C.to_mont(C.FIELD_FQ, np.array([1, 0, 0, 0], dtype=np.uint64))[0]     # a heavy bucket
    assert (key.commit(sc) == C.commit(cid, bs, sc)).all()
    key.precompute()
    d = emu_lib.alloc(n * 32); emu_lib.upload(d, sc)
    # ONE launch over the tables (2^19 emulated buckets take half a minute): the chunk [90, n) of the key, table rows
    # offset by `first`, against the oracle's commit of that chunk
    pb, cb, wb = key.commit_partial_device(90, d + 90 * 32, n - 90)
    assert (cb, wb) == (0, 64)
    assert (cm.combine_partials(cid, np.stack([pb]), cb, wb, lib=emu_lib) == C.commit(cid, bs[90:], sc[90:])).all()


def test_emu_shared_bucket_tables_16bit(emu_lib, tune):
    """mira_msm_precompute_ex(handle, 16): tables 2^(16 w) P_i, ONE set of 2^15 buckets for the 16
    windows through the per-window launch sequence, 16 partial sums back.  Same points as the per-window path and the
    oracle; identity base, heavy bucket, chunk partials, a batch.  (2^15 emulated buckets are slow: the staged sort, host
    scalars in chunks and prefixes of the shared-bucket path run at narrower widths in the next test.)"""
    tune(_lib.TUNE_TABLE_MIN_N, 1)
    tune(_lib.TUNE_SHARED_MIN_N, 1)
    cid, n = 0, 300
    bs = C.synth_bases(cid, n, seed=44)
    bs[11] = 0
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    sc = C.synth_scalars(cid, n, seed=45, kind=1)
    sc[:50] = C.to_mont(C.FIELD_FR, np.array([1, 0, 0, 0], dtype=np.uint64))[0]        # a heavy bucket
    dense = C.synth_scalars(cid, n, seed=46)
    want, want_dense = C.commit(cid, bs, sc), C.commit(cid, bs, dense)
    key.precompute(16)
    assert (key.commit(sc) == want).all()
    d = emu_lib.alloc(n * 32); emu_lib.upload(d, dense)
    pa, ca, wa = key.commit_partial_device(0, d, 130)
    pb, cb, wb = key.commit_partial_device(130, d + 130 * 32, n - 130)
    assert (ca, wa) == (0, 1) == (cb, wb)
    assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa, lib=emu_lib) == want_dense).all()
    # a batch over the tables: one bucket set per commitment (two vectors of a prefix length, one all zeros)
    vs = [dense[:200], np.zeros((200, 4), dtype=np.uint64)]
    want_b = np.stack([C.commit(cid, bs[:200], v) for v in vs])
    assert (key.commit_batch(vs) == want_b).all()
    with pytest.raises(_lib.MiraError):
        key.precompute(18)                                    # shared buckets are 8 .. 16 bits, wide tables 20 or 22


def test_emu_shared_bucket_tables_other_widths(emu_lib, tune):
    """mira_msm_precompute_ex(handle, c) for c = 8 .. 15 (here 8 and 13): W = ceil(256 / c) tables, ONE set of 2^(c-1) buckets,
    min(16, 2^(c-3)) partial sums back.  Several widths live beside each other on one key; MIRA_TUNE_TABLE_WIDTH names
    the set a commit goes through, mira_msm_last_table_bits reports it; without the knob the commit's length picks
    one.  Same points as the per-window path and the oracle for single commits, prefixes, a batch, chunk partials
    (ranks of a sharded MSM all take the widest set) and host scalars in point chunks."""
    tune(_lib.TUNE_TABLE_MIN_N, 1)
    tune(_lib.TUNE_SHARED_MIN_N, 1)
    cid, n = 1, 260
    bs = C.synth_bases(cid, n, seed=47)
    bs[5] = 0
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    sc = C.synth_scalars(cid, n, seed=48, kind=1)
    sc[:60] = C.to_mont(C.FIELD_FQ, np.array([1, 0, 0, 0], dtype=np.uint64))[0]        # a heavy bucket
    dense = C.synth_scalars(cid, n, seed=49)
    want, want_dense = C.commit(cid, bs, sc), C.commit(cid, bs, dense)
    tb = ctypes.c_int32()

    def last_table():
        emu_lib.check(emu_lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
        return tb.value
    assert (key.commit(dense) == want_dense).all() and last_table() == 0
    vs = [dense[:150], sc[:150]]
    want_b = np.stack([C.commit(cid, bs[:150], v) for v in vs])
    for c in (8, 13):
        key.precompute(c)
        key.precompute(c)                                     # a second build of the same width is a no-op
        tune(_lib.TUNE_TABLE_WIDTH, c)
        assert (key.commit(sc) == want).all() and last_table() == c
        if c == 8:
            assert (key.commit(dense[:77]) == C.commit(cid, bs[:77], dense[:77])).all()
        assert (key.commit_batch(vs) == want_b).all() and last_table() == c
    tune(_lib.TUNE_TABLE_WIDTH, 8)
    d = emu_lib.alloc(n * 32); emu_lib.upload(d, dense)
    pa, ca, wa = key.commit_partial_device(0, d, 100)
    pb, cb, wb = key.commit_partial_device(100, d + 100 * 32, n - 100)
    assert (ca, wa) == (0, 1) == (cb, wb) and last_table() == 8
    assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa, lib=emu_lib) == want_dense).all()
    tune(_lib.TUNE_STAGED_MIN_N, 1)                           # the LDS-staged sort with table indices, 12 fine bits at most
    tune(_lib.TUNE_TABLE_WIDTH, 13)
    assert (key.commit_batch(vs) == want_b).all()           # (set 0 of the batch is a lone commit's case)
    tune(_lib.TUNE_HOST_CHUNK_MIN_N, 64)                      # host scalars in chunks of 32, 64, 128, ... points
    assert (key.commit(sc) == want).all()
    tune(_lib.TUNE_TABLE_WIDTH, -1)                           # the length picks a set
    assert (key.commit(dense) == want_dense).all() and last_table() in (8, 13)
    # ... and, from the second commit of a shape on, the bit lengths of the previous one do (they never change a result)
    tune(_lib.TUNE_PLAN_HIST_MIN_N, 1)
    for v, w in ((sc, want), (dense, want_dense), (sc, want)):
        assert (key.commit(v) == w).all() and last_table() in (8, 13)


@pytest.mark.parametrize("cid", [0, 1])
def test_emu_glv_split(emu_lib, tune, cid):
    """mira_msm_precompute_ex(handle, MIRA_TABLE_GLV): every scalar split into two signed 127-bit halves over the interleaved key
    [P_i, phi(P_i)] (glv.cuh).  Same points as the oracle for scalars around the decomposition's edges (0, 1, 2^127 -+ 1, 2^128,
    r - 1, r - 2, 2^253), an identity base, dense and witness-like vectors, planned and forced widths (5 .. 16: ceil(128 / c)
    windows), a prefix, a batch, host scalars in point chunks, and with the statistics of the previous commit planning the next;
    switched off by MIRA_TUNE_GLV = 0; chunk partials of a sharded MSM keep the plain shape."""
    n = 300
    r = P.CURVES[cid].r
    bs = C.synth_bases(cid, n, seed=91)
    bs[9] = 0
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    key.precompute(_lib.TABLE_GLV)
    key.precompute(_lib.TABLE_GLV)                            # twice: a no-op
    edge = [0, 1, 2, (1 << 127) - 1, 1 << 127, (1 << 127) + 1, 1 << 128, r - 1, r - 2, 1 << 253, (1 << 253) + 12345]
    cc, ww = ctypes.c_int32(), ctypes.c_int32()

    def last_plan():
        emu_lib.check(emu_lib.c.mira_msm_last_plan(ctypes.byref(cc), ctypes.byref(ww)))
        return cc.value, ww.value
    for kind, seed in ((0, 1), (1, 2)):
        sc = C.synth_scalars(cid, n, seed=seed, kind=kind)
        sc[20:20 + len(edge)] = ints_to_mont(edge, r)
        want = C.commit(cid, bs, sc)
        assert (key.commit(sc) == want).all()
        c0, w0 = last_plan()
        assert w0 == -(-128 // c0)                            # half the windows of the plain path
        for c in (((16,) if cid == 0 else (9,)) if kind == 0 else ()):   # (every emulated commit takes seconds: the GPU suite sweeps the widths)
            emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
            try:
                assert (key.commit(sc) == want).all(), c
                assert last_plan() == (c, -(-128 // c))
            finally:
                emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))
    dense = C.synth_scalars(cid, n, seed=7)
    want_dense = C.commit(cid, bs, dense)
    assert (key.commit(dense[:77]) == C.commit(cid, bs[:77], dense[:77])).all()
    vs = [dense[:150], sc[:150], np.zeros((150, 4), dtype=np.uint64)]          # a batch: 2 x 150 halves per commitment
    assert (key.commit_batch(vs) == np.stack([C.commit(cid, bs[:150], v) for v in vs])).all() and last_plan()[1] == -(-128 // last_plan()[0])
    tune(_lib.TUNE_GLV, 0)
    assert (key.commit(dense) == want_dense).all() and last_plan()[1] == -(-256 // last_plan()[0])
    tune(_lib.TUNE_GLV, 1)
    tune(_lib.TUNE_HOST_CHUNK_MIN_N, 64)                      # host scalars in chunks of 32, 64, 128, ... points: rows 2 lo .. of the interleaved key
    assert (key.commit(dense) == want_dense).all()
    tune(_lib.TUNE_PLAN_HIST_MIN_N, 1)                        # the bit lengths of the halves of one commit plan the next
    d = emu_lib.alloc(n * 32)
    for v, w in ((sc, want), (dense, want_dense)):
        emu_lib.upload(d, v)
        assert (key.commit_device(d, n) == w).all()
    pb, cb, wb = key.commit_partial_device(120, d + 120 * 32, n - 120)
    assert wb == -(-256 // cb)
    assert (cm.combine_partials(cid, np.stack([pb]), cb, wb, lib=emu_lib) == C.commit(cid, bs[120:], dense[120:])).all()
    emu_lib.free(d)


def test_emu_data_dependent_planning(emu_lib, tune):
    """The bit-length statistics of one commit plan the next one of the same length over the key:
    they change only the window width, never the result."""
    tune(_lib.TUNE_PLAN_HIST_MIN_N, 1)
    cid, n = 0, 300
    bs = C.synth_bases(cid, n, seed=50)
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    sc = C.synth_scalars(cid, n, seed=52, kind=1)
    want = C.commit(cid, bs, sc)
    assert (key.commit(sc) == want).all()                     # planned from the dense estimate
    assert (key.commit(sc) == want).all()                     # planned from the first call's histogram
    assert not key.commit(np.zeros((n, 4), dtype=np.uint64)).any()
    assert not key.commit(np.zeros((n, 4), dtype=np.uint64)).any()   # planned from an all-zero histogram
    assert (key.commit(sc) == want).all()
    # mira_msm_last_plan: the planner's own choice, then a forced width
    import ctypes
    c, w = ctypes.c_int32(), ctypes.c_int32()
    emu_lib.check(emu_lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
    assert 4 <= c.value <= 16 and w.value == -(-256 // c.value)
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(11))
    try:
        assert (key.commit(sc) == want).all()
        emu_lib.check(emu_lib.c.mira_msm_last_plan(ctypes.byref(c), ctypes.byref(w)))
        assert (c.value, w.value) == (11, 24)
    finally:
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))


def test_emu_staged_sort(emu_lib, tune):
    """LDS-staged two-level sort (sort_kernels.cuh), forced on at a small size: same commitments."""
    tune(_lib.TUNE_STAGED_MIN_N, 1)
    for cid, c in ((1, 11),):
        n = 500
        bs = C.synth_bases(cid, n, seed=60)
        key = cm.CommitmentKey(cid, bs, lib=emu_lib)
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
        try:
            sc = C.synth_scalars(cid, n, seed=62, kind=1)
            assert (key.commit(sc) == C.commit(cid, bs, sc)).all()
            one = np.tile(C.to_mont(C.FIELD_FR if cid == 0 else C.FIELD_FQ, np.array([1, 0, 0, 0], dtype=np.uint64)), (n, 1))
            assert (key.commit(one) == C.msm_pippenger(cid, one, bs)).all()
        finally:
            emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))


def test_emu_staged_sort_full_tiles(emu_lib, tune):
    """Whole 8192-entry sort tiles beside a ragged one, and k_hist's four-digits-per-load path (taken
    where a window's digits are 8-byte aligned: n is odd, so only every fourth window is)."""
    tune(_lib.TUNE_STAGED_MIN_N, 1)
    cid, n, c = 0, 8192 + 77, 12
    key = cm.CommitmentKey.synthetic(cid, n, seed=66, lib=emu_lib)
    bs = key.download()
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
    try:
        sc = C.synth_scalars(cid, n, seed=67, kind=1)
        assert (key.commit(sc) == C.commit(cid, bs, sc)).all()
    finally:
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))


def test_emu_host_scalars_in_chunks(emu_lib, tune):
    """Host scalars cut into point chunks (copy of one chunk beside the kernels of the previous one on
    the GPU): every chunk adds its bucket sums to those before.  Same commitments as the one-chunk
    path, with heavy buckets, zeros and an identity base spread over the chunks."""
    tune(_lib.TUNE_HOST_CHUNK_MIN_N, 64)                     # chunks of 32, 64, 128, 128, ... points
    for cid, c in ((0, 9), (1, 5)):
        n = 700
        bs = C.synth_bases(cid, n, seed=70 + cid)
        bs[300] = 0
        key = cm.CommitmentKey(cid, bs, lib=emu_lib)
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(c))
        try:
            for kind in (0, 1):
                sc = C.synth_scalars(cid, n, seed=72 + kind, kind=kind)
                sc[100:160] = sc[0]                            # one bucket per window fed from several chunks
                assert (key.commit(sc) == C.commit(cid, bs, sc)).all(), (cid, kind)
            assert (key.commit(sc[:33]) == C.commit(cid, bs[:33], sc[:33])).all()      # below the threshold: one chunk
        finally:
            emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))


def test_emu_endomorphism_copy_built_at_first_use(emu_lib, tune):
    """MIRA_TUNE_GLV_AUTO_MAX_LOG: a key of at least 2^12 points gets its endomorphism copy the first time a commit can use it --
    no precompute call; the commit then has ceil(128 / c) windows and the oracle's point.  Switched off (0), or for a key longer
    than the limit, the plain path stays; ranks of a sharded MSM never take it."""
    cid, n, m = 1, 1 << 12, 180
    key = cm.CommitmentKey.synthetic(cid, n, seed=97, lib=emu_lib)
    bs = key.download(0, m)
    sc = C.synth_scalars(cid, m, seed=98)
    want = C.commit(cid, bs, sc)
    cc, ww = ctypes.c_int32(), ctypes.c_int32()

    def windows():
        emu_lib.check(emu_lib.c.mira_msm_last_plan(ctypes.byref(cc), ctypes.byref(ww)))
        return ww.value * cc.value
    # (a FORCED width takes the split wherever the key may have its copy; a planned commit asks both planners and takes the split
    # only where it is estimated ahead -- from ~2^12 to ~2^19 pairs, sizes the emulation does not reach)
    key.set_window_bits(9)
    tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 0)
    assert (key.commit(sc) == want).all() and windows() >= 254          # no copy, no split
    tune(_lib.TUNE_GLV_AUTO_MAX_LOG, 11)                                # the key is longer than 2^11 points: still none
    assert (key.commit(sc) == want).all() and windows() >= 254
    tune(_lib.TUNE_GLV_AUTO_MAX_LOG, -1)                                # default: built now, used from now on
    assert (key.commit(sc) == want).all() and 127 <= windows() < 160
    assert (key.commit_batch([sc, sc[::-1].copy()]) == np.stack([want, C.commit(cid, bs, sc[::-1].copy())])).all() and 127 <= windows() < 160
    key.set_window_bits(0)
    assert (key.commit(sc) == want).all() and (windows() >= 254 or 127 <= windows() < 160)   # planned: whichever path the two tables estimate ahead (re-measured at the end of round 4: the split, from 2^10 pairs down)
    d = emu_lib.alloc(m * 32); emu_lib.upload(d, sc)
    part, c, w = key.commit_partial_device(0, d, m, window_bits=9)      # a rank of a sharded MSM: the plain shape
    assert (c, w) == (9, 29) and (cm.combine_partials(cid, part[None, :], c, w, lib=emu_lib) == want).all()
    emu_lib.free(d); key.close()


def test_emu_long_commits_and_host_batches_in_chunks(emu_lib, tune):
    """Two more users of the chunk path of msm_host.cuh: a commit of more sorted entries than one pass takes (n W >= 2^32 on
    the GPU: the reference's 2^27 .. 2^28-point keys; here the pass is shrunk by MIRA_TUNE_PASS_ENTRIES_LOG) is cut into point
    chunks that add into one set of buckets -- scalars in device memory, a prefix, a batch, the GLV copy and a shared-bucket
    set -- and the vectors of a HOST batch (mira_msm_batch) cross in point chunks, staged slice by slice."""
    cid, n = 0, 420
    bs = C.synth_bases(cid, n, seed=90)
    bs[300] = 0
    key = cm.CommitmentKey(cid, bs, lib=emu_lib)
    vs = [C.synth_scalars(cid, n, seed=91 + b, kind=b % 2) for b in range(2)]
    for v in vs:
        v[150:200] = v[1]
    want = [C.commit(cid, bs, v) for v in vs]
    d = emu_lib.alloc(2 * n * 32)
    for b, v in enumerate(vs):
        emu_lib.upload(d + b * n * 32, v)
    emu_lib.check(emu_lib.c.mira_msm_set_window_bits(9))
    try:
        tune(_lib.TUNE_PASS_ENTRIES_LOG, 13)                     # 8191 entries per pass: chunks of 256 points under 29 windows of 9 bits
        assert (key.commit_device(d, n) == want[0]).all()
        assert (key.commit_device(d + n * 32, 333) == C.commit(cid, bs[:333], vs[1][:333])).all()
        assert (key.commit_batch_device(d, n, 2) == np.stack(want)).all()      # a batch is cut by count first, then by points
        tune(_lib.TUNE_HOST_CHUNK_MIN_N, 256)                    # host vectors: the copy chunks (128, 256 ... scalars) are cut further by the pass
        assert (key.commit(vs[1]) == want[1]).all()
        assert (key.commit_batch(vs) == np.stack(want)).all()    # a host batch: slices of every vector per chunk
        tune(_lib.TUNE_PASS_ENTRIES_LOG, -1)
        assert (key.commit_batch(vs) == np.stack(want)).all()
        assert (key.commit_batch([v[:50] for v in vs]) == np.stack([C.commit(cid, bs[:50], v[:50]) for v in vs])).all()   # below the threshold: one chunk
    finally:
        emu_lib.check(emu_lib.c.mira_msm_set_window_bits(0))
    tune(_lib.TUNE_PASS_ENTRIES_LOG, 13)
    key.precompute(_lib.TABLE_GLV)                               # the endomorphism copy: 2 n columns per chunk
    key.set_window_bits(9)
    assert (key.commit_device(d, n) == want[0]).all()
    key.set_window_bits(0)
    tune(_lib.TUNE_SHARED_MIN_N, 1)
    key.precompute(9)
    assert (key.commit_device(d, n) == want[0]).all() and (key.commit_batch(vs) == np.stack(want)).all()
    emu_lib.free(d); key.close()


def test_emu_medium_runs_both_placements(emu_lib):
    """Runs cut into 7 .. 12 partials are placed on the device once their number is known (msm_kernels.cuh: MEDIUM_SPAN): MANY of
    them BESIDE a really heavy run are summed as chains by the quads of the light section (6-bit windows over 3 000 dense scalars:
    every bucket holds ~ 90 entries, nine segments of ten; 400 copies of one small value make one bucket heavy), many of them alone
    or a FEW become sub-jobs of the heavy section (one value repeated 90 times under 9-bit windows:
    one such run beside light ones).  The emulation build switches between the two at 8 runs."""
    cid = 0
    for n, c, dup in ((3000, 6, 400), (3000, 6, 0), (300, 9, 90)):     # many medium runs beside a really heavy one: chains; many alone, or a few: sub-jobs
        bs = C.synth_bases(cid, n, seed=101 + c)
        sc = C.synth_scalars(cid, n, seed=102 + c)
        if dup:
            sc[100:100 + dup] = ints_to_mont([5], P.CURVES[cid].r)[0]
        key = cm.CommitmentKey(cid, bs, lib=emu_lib)
        key.set_window_bits(c)
        assert (key.commit(sc) == C.msm_pippenger(cid, sc, bs)).all(), (n, c)
        key.close()


@pytest.mark.parametrize("n,c", [(20000, 4)])
def test_emu_heavy_subjob_grouping(emu_lib, n, c):
    """k_fixup_heavy_a gives a sub-job 16, 8, 4 or 2 quads by the number of sub-jobs: 4-bit windows over 20 000 dense scalars
    make every bucket a heavy run of three sub-jobs -- 1 536 sub-jobs, 8 quads each (4 and 2 quads per sub-job: the -m gpu
    suite, tests/test_gpu_msm.py::test_every_bucket_heavy)."""
    cid = 1
    key = cm.CommitmentKey.synthetic(cid, n, seed=17, lib=emu_lib)
    key.set_window_bits(c)
    d = cm.synth_scalars_device(cid, n, seed=18, lib=emu_lib)
    sc = emu_lib.download(d, (n, 4))
    assert (key.commit_device(d, n) == C.msm_pippenger(cid, sc, key.download())).all()
    emu_lib.free(d); key.close()
