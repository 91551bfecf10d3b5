"""GPU parity tests for CommitmentKey::commit (reference src/commitment.rs:78-87): the HIP path
through the C ABI against the CPU oracle on identical inputs -- bit-exact (integer work)."""
import ctypes

import numpy as np
import pytest

from helpers import arr_to_point, golden_msm_case, ints_to_mont, load_golden, mont_to_ints, point_to_arr
from mira_amd import commitment as cm
from mira_amd import _lib
from oracle import cref as C
from oracle import pyref as P

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["endomorphism copy at first use", "plain keys"])
def glv_auto(request, gpu_lib):
    """Every test of this file runs twice: with the library's default -- a key of >= 2^12 points gets its endomorphism copy at the
    first commit of <= 2^21 pairs, which then takes the GLV split (MIRA_TUNE_GLV_AUTO_MAX_LOG) -- and with that switched off, so
    that the plain per-window path keeps the coverage it had before the copy became automatic."""
    gpu_lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, -1 if request.param.startswith("endo") else 0)
    yield request.param
    gpu_lib.tune(_lib.TUNE_GLV_AUTO_MAX_LOG, -1)


@pytest.mark.parametrize("cid", [0, 1])
def test_synth_generators(gpu_lib, cid):
    n = 3000
    for kind in (0, 1):
        p = cm.synth_scalars_device(cid, n, seed=31, kind=kind)
        assert (gpu_lib.download(p, (n, 4)) == C.synth_scalars(cid, n, seed=31, kind=kind)).all()
        gpu_lib.free(p)
    key = cm.CommitmentKey.synthetic(cid, n, seed=32)
    assert (key.bases() == C.synth_bases(cid, n, seed=32)).all()
    key.check_on_curve()


@pytest.mark.parametrize("cid", [0, 1])
def test_golden_vectors(gpu_lib, cid):
    for case in load_golden("msm_vectors.json")[str(cid)]:
        sc, bs, expected = golden_msm_case(cid, case)
        key = cm.CommitmentKey(cid, bs)
        assert (key.commit(sc) == expected).all(), case["n"]
        for c in (8, 13, 16):       # every window width gives the same group element
            gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(c))
            try:
                assert (key.commit(sc) == expected).all(), (case["n"], c)
            finally:
                gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))


def test_g1_scalar_mul_kat(gpu_lib):
    # src/digest.rs:98-113: (r-1) * G == -G
    g = C.generator(0)
    key = cm.CommitmentKey(0, g.reshape(1, 8))
    k = ints_to_mont([P.R_MOD - 1], P.R_MOD)
    assert arr_to_point(key.commit(k), 0) == (1, P.P_MOD - 2)


@pytest.mark.parametrize("cid,n,kind", [(0, 1 << 16, 0), (0, 1 << 16, 1), (1, 1 << 16, 0), (1, 50001, 1), (0, 131072, 0), (0, 1 << 18, 0)])
def test_parity_vs_oracle(gpu_lib, cid, n, kind):
    """BASELINE config 0/1 shape: random scalars/points via commit, GPU vs CPU oracle."""
    key = cm.CommitmentKey.synthetic(cid, n, seed=41)
    bases = key.bases()
    sc = C.synth_scalars(cid, n, seed=42, kind=kind)
    assert (key.commit(sc) == C.commit(cid, bases, sc)).all()


def test_edge_cases(gpu_lib):
    cid = 0
    n = 5000
    bs = C.synth_bases(cid, n, seed=7)
    key = cm.CommitmentKey(cid, bs)
    zero = np.zeros((n, 4), dtype=np.uint64)
    assert not key.commit(zero).any()                                   # all-zero scalars -> identity
    assert not key.commit(np.zeros((0, 4), dtype=np.uint64)).any()      # empty -> identity
    one = np.tile(C.to_mont(C.FIELD_FR, np.array([1, 0, 0, 0], dtype=np.uint64)), (n, 1))
    assert (key.commit(one) == C.msm_pippenger(cid, one, bs)).all()      # one heavy bucket
    rm1 = np.tile(ints_to_mont([P.R_MOD - 1], P.R_MOD), (n, 1))
    assert (key.commit(rm1) == C.msm_pippenger(cid, rm1, bs)).all()      # top digits / carries
    same = np.tile(bs[:1], (n, 1))                                        # one base repeated: doubling path
    skey = cm.CommitmentKey(cid, same)
    sc = C.synth_scalars(cid, n, seed=9)
    assert (skey.commit(sc) == C.msm_pippenger(cid, sc, same)).all()
    ident = np.zeros((n, 8), dtype=np.uint64)                             # identity bases
    assert not cm.CommitmentKey(cid, ident).commit(sc).any()
    # (negation handled in the golden vectors; here check prefix semantics and TooLongInput)
    assert (key.commit(sc[:1234]) == C.msm_pippenger(cid, sc[:1234], bs[:1234])).all()
    with pytest.raises(cm.TooLongInput):
        key.commit(np.zeros((n + 1, 4), dtype=np.uint64))


def test_witness_like_heavy_buckets(gpu_lib):
    """Column-major witness vector (src/util.rs:189-193) with tiny repeated values: a few buckets
    hold almost everything."""
    cid, rows, cols = 0, 1 << 14, 5
    n = rows * cols
    key = cm.CommitmentKey.synthetic(cid, n, seed=51)
    bases = key.bases()
    rng = np.random.default_rng(1)
    small = rng.integers(0, 3, size=n).astype(object)            # values 0, 1, 2
    sc = ints_to_mont([int(v) for v in small], P.R_MOD)
    assert (key.commit(sc) == C.commit(cid, bases, sc)).all()
    padded = cm.concatenate_with_padding([sc[:1000], sc[1000:1500]], 2048)
    assert (key.commit(padded) == C.commit(cid, bases, padded)).all()


def test_homomorphism_prefix(gpu_lib):
    """Com(W1 + r W2) == Com(W1) + r Com(W2) on affine points (src/plonk/mod.rs:547-557,
    src/nifs/vanilla/tests.rs:189,228), GPU commits, EC arithmetic of the check in Python ints."""
    for cid in (0, 1):
        cv = P.CURVES[cid]
        n = 4096
        key = cm.CommitmentKey.synthetic(cid, n, seed=61)
        w1 = C.synth_scalars(cid, n, seed=62, kind=1)
        w2 = C.synth_scalars(cid, n, seed=63)
        a, b = mont_to_ints(w1, cv.r), mont_to_ints(w2, cv.r)
        r = P.synth_scalar(5, cv.r)
        folded = ints_to_mont([(x + r * y) % cv.r for x, y in zip(a, b)], cv.r)
        p1, p2 = arr_to_point(key.commit(w1), cid), arr_to_point(key.commit(w2), cid)
        assert arr_to_point(key.commit(folded), cid) == P.ec_add(p1, P.ec_mul(r, p2, cv), cv)


@pytest.mark.parametrize("log_n", [20, 22])
def test_full_size_properties_and_parity(gpu_lib, log_n):
    """BASELINE config 1 (2^22, 16-bit windows) and 2^20: bit-exact vs the oracle, and
    size-independent properties: every window width gives the same point; chunk partials
    combine to the whole."""
    cid, n = 0, 1 << log_n
    key = cm.CommitmentKey.synthetic(cid, n, seed=81)
    d = cm.synth_scalars_device(cid, n, seed=82)
    whole = key.commit_device(d, n)
    assert whole.any()
    for c in (13, 16):
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(c))
        try:
            assert (key.commit_device(d, n) == whole).all()
            h = n // 3
            pa, ca, wa = key.commit_partial_device(0, d, h)
            pb, cb, wb = key.commit_partial_device(h, d + h * 32, n - h)
            assert (ca, wa) == (cb, wb)
            assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa) == whole).all()
        finally:
            gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
    sc = gpu_lib.download(d, (n, 4))
    assert (whole == C.commit(cid, key.bases(), sc)).all()


def test_sharded_partials_equal_single(gpu_lib):
    """BASELINE config 4 shape on one GPU: G chunk partials combined == one MSM."""
    cid, n, G = 0, 1 << 18, 8
    key = cm.CommitmentKey.synthetic(cid, n, seed=71)
    d = cm.synth_scalars_device(cid, n, seed=72)
    whole = key.commit_device(d, n)
    gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(16))
    try:
        parts = []
        per = n // G
        for g in range(G):
            part, c, w = key.commit_partial_device(g * per, d + g * per * 32, per)
            parts.append(part)
        assert (cm.combine_partials(cid, np.stack(parts), c, w) == whole).all()
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))


@pytest.mark.parametrize("cid,count", [(0, 6), (1, 5)])
def test_commit_batch_fold_step_shape(gpu_lib, cid, count):
    """The cross-term commits of one k = 17 fold step as one batched submission: every point equals
    the oracle's and the one-at-a-time commit."""
    n = 1 << 17
    key = cm.CommitmentKey.synthetic(cid, n, seed=91)
    bases = key.bases()
    vs = [C.synth_scalars(cid, n, seed=100 + i) for i in range(count)]
    got = key.commit_batch(vs)
    for i, v in enumerate(vs):
        assert (got[i] == C.commit(cid, bases, v)).all()
    assert (got[0] == key.commit(vs[0])).all()
    # device form with a stride larger than n
    d = gpu_lib.alloc(count * (n + 64) * 32)
    for i, v in enumerate(vs):
        gpu_lib.upload(d + i * (n + 64) * 32, v)
    assert (key.commit_batch_device(d, n, count, stride=n + 64) == got).all()
    gpu_lib.free(d)


def test_commit_batch_chunking(gpu_lib):
    """More vectors than one launch takes at 16-bit windows (scan capacity): chunked internally."""
    cid, n, count = 0, 1 << 14, 9
    key = cm.CommitmentKey.synthetic(cid, n, seed=92)
    bases = key.bases()
    vs = [C.synth_scalars(cid, n, seed=200 + i, kind=i % 2) for i in range(count)]
    gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(16))
    try:
        got = key.commit_batch(vs)
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
    for i, v in enumerate(vs):
        assert (got[i] == C.commit(cid, bases, v)).all()


@pytest.mark.parametrize("cid,log_n,width", [(0, 20, 20), (1, 19, 20), (0, 20, 22), (1, 19, 22)])
def test_fixed_base_tables(gpu_lib, cid, log_n, width):
    """mira_msm_precompute (window tables in HBM, 2^19 shared buckets): bit-identical to the
    per-window path and the oracle; partials over the tables combine to the whole."""
    n = 1 << log_n
    key = cm.CommitmentKey.synthetic(cid, n, seed=93)
    d = cm.synth_scalars_device(cid, n, seed=94)
    dw = cm.synth_scalars_device(cid, n, seed=95, kind=1)
    before, before_w = key.commit_device(d, n), key.commit_device(dw, n)
    key.precompute(width)                                 # 13 tables / 2^19 buckets, or 12 tables / 2^21 buckets
    assert (key.commit_device(d, n) == before).all()
    assert (key.commit_device(dw, n) == before_w).all()
    assert (before == C.commit(cid, key.bases(), gpu_lib.download(d, (n, 4)))).all()
    h = n // 2 + 12345
    pa, ca, wa = key.commit_partial_device(0, d, h)
    pb, cb, wb = key.commit_partial_device(h, d + h * 32, n - h)
    assert (ca, wa) == (0, 64) == (cb, wb)               # partials of a key with tables are always table-mode
    assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa) == before).all()
    m = (1 << 18) + 777                                  # a prefix of the key, still table mode
    assert (key.commit_device(d, m) == C.commit(cid, key.bases()[:m], gpu_lib.download(d, (m, 4)))).all()


@pytest.mark.parametrize("cid", [0, 1])
def test_staged_sort_under_skew(gpu_lib, cid):
    """The LDS-staged sort (n >= 2^19) on inputs that break its assumptions: witness-like vectors
    (sparse: level-2 tiles span many coarse bins and take the direct-placement path), one giant
    bucket, and a column-major mix -- per-window path and fixed-base tables, all against the oracle."""
    n = 1 << 20
    key = cm.CommitmentKey.synthetic(cid, n, seed=97)
    bases = key.bases()
    field = C.FIELD_FR if cid == 0 else C.FIELD_FQ
    one = C.to_mont(field, np.array([1, 0, 0, 0], dtype=np.uint64))[0]
    wit = C.synth_scalars(cid, n, seed=98, kind=1)
    ones = np.tile(one, (n, 1))
    mix = C.synth_scalars(cid, n, seed=99)
    mix[: n // 2] = one                                   # half the vector in one bucket, half uniform
    mix[n // 2: n // 2 + n // 8] = 0
    cases = [wit, ones, mix]
    want = [C.commit(cid, bases, v) for v in cases]
    for c in (0, 16):                                      # planned width (bit-length pre-pass) and forced 16
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(c))
        try:
            for v, w in zip(cases, want):
                assert (key.commit(v) == w).all()
        finally:
            gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
    key.precompute()
    for v, w in zip(cases, want):
        assert (key.commit(v) == w).all()


def test_abi_from_two_threads(gpu_lib):
    """Two caller threads, two keys, interleaved mira_msm + mira_fft_bn256_fr (include/mira_gpu.h: re-entrant)."""
    from test_host_logic import _two_thread_abi
    _two_thread_abi(gpu_lib, n=20000, log_n=14, rounds=6)


@pytest.mark.parametrize("cid,kind", [(0, 0), (1, 1)])
def test_host_scalars_cross_pcie_in_chunks(gpu_lib, cid, kind):
    """commit(v) with v in host memory (src/commitment.rs:78) above 2^19 pairs: point chunks whose
    copies overlap the kernels of the previous chunk, bucket sums added across chunks.  Same point
    as the device-resident commit and the oracle; a forced small threshold gives many chunks."""
    from mira_amd import _lib
    n = (1 << 20) + 12345
    key = cm.CommitmentKey.synthetic(cid, n, seed=111)
    sc = C.synth_scalars(cid, n, seed=112, kind=kind)
    d = gpu_lib.alloc(n * 32); gpu_lib.upload(d, sc)
    want = key.commit_device(d, n)
    assert (key.commit(sc) == want).all()
    assert (want == C.commit(cid, key.bases(), sc)).all()
    gpu_lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, 1 << 14)        # 8k, 16k, 32k, 32k, ... pairs per chunk
    try:
        assert (key.commit(sc) == want).all()
        assert (key.commit(sc[: 1 << 15]) == key.commit_device(d, 1 << 15)).all()
    finally:
        gpu_lib.tune(_lib.TUNE_HOST_CHUNK_MIN_N, -1)
    gpu_lib.free(d)


@pytest.mark.parametrize("cid,log_n", [(0, 17), (1, 20)])
def test_shared_bucket_tables_16bit(gpu_lib, cid, log_n):
    """mira_msm_precompute_ex(handle, 16): one set of 2^15 buckets for all 16 windows.  Bit-identical
    to the per-window path and the oracle: dense and witness-like vectors, a prefix, chunk partials,
    host scalars (in point chunks at 2^20)."""
    n = 1 << log_n
    key = cm.CommitmentKey.synthetic(cid, n, seed=121)
    d = cm.synth_scalars_device(cid, n, seed=122)
    dw = cm.synth_scalars_device(cid, n, seed=123, kind=1)
    before, before_w = key.commit_device(d, n), key.commit_device(dw, n)
    key.precompute(16)
    assert (key.commit_device(d, n) == before).all()
    assert (key.commit_device(dw, n) == before_w).all()
    sc = gpu_lib.download(d, (n, 4))
    assert (before == C.commit(cid, key.bases(), sc)).all()
    assert (key.commit(sc) == before).all()                                  # host scalars
    m = n // 2 + 4321
    assert (key.commit_device(d, m) == C.commit(cid, key.bases()[:m], sc[:m])).all()
    pa, ca, wa = key.commit_partial_device(0, d, m)
    pb, cb, wb = key.commit_partial_device(m, d + m * 32, n - m)
    assert (ca, wa) == (0, 1) == (cb, wb)
    assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa) == before).all()
    # a batch over the tables (one bucket set per commitment): five cross-term-sized vectors at a stride
    nb, cnt = 1 << 15, 5
    want_b = np.stack([C.commit(cid, key.bases()[:nb], sc[b * 3 * nb // 2: b * 3 * nb // 2 + nb]) for b in range(cnt)]) if log_n >= 20 else None
    got_b = key.commit_batch_device(d, nb if log_n >= 20 else n // 8, cnt, stride=3 * nb // 2 if log_n >= 20 else n // 8)
    if want_b is not None:
        assert (got_b == want_b).all()
    else:
        assert all((got_b[b] == key.commit_device(d + b * (n // 8) * 32, n // 8)).all() for b in range(cnt))
    gpu_lib.free(d); gpu_lib.free(dw)


@pytest.mark.parametrize("cid,log_n", [(0, 17), (1, 19)])
def test_shared_bucket_tables_every_width(gpu_lib, cid, log_n):
    """mira_msm_precompute_ex(handle, c), c = 8 .. 16, all nine sets beside each other on one key: every width gives the
    per-window path's point (which the oracle confirms) for a dense and a witness-like vector, a prefix, host scalars and
    a batch of five; chunk partials of a sharded MSM take the widest set; without MIRA_TUNE_TABLE_WIDTH the commit's
    length picks a set (mira_msm_last_table_bits)."""
    n = 1 << log_n
    key = cm.CommitmentKey.synthetic(cid, n, seed=131)
    d = cm.synth_scalars_device(cid, n, seed=132)
    dw = cm.synth_scalars_device(cid, n, seed=133, kind=1)
    before, before_w = key.commit_device(d, n), key.commit_device(dw, n)
    sc = gpu_lib.download(d, (n, 4))
    assert (before == C.commit(cid, key.bases(), sc)).all()
    m = n // 2 + 1234
    before_m = key.commit_device(d, m)
    nb, cnt = n // 8, 5
    before_b = key.commit_batch_device(d, nb, cnt, stride=nb + 7)
    tb = ctypes.c_int32()
    try:
        for c in range(8, 17):
            key.precompute(c)
            gpu_lib.tune(_lib.TUNE_TABLE_WIDTH, c)
            assert (key.commit_device(d, n) == before).all(), c
            gpu_lib.check(gpu_lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
            assert tb.value == c
            assert (key.commit_device(dw, n) == before_w).all(), c
            assert (key.commit_device(d, m) == before_m).all(), c
            assert (key.commit_batch_device(d, nb, cnt, stride=nb + 7) == before_b).all(), c
            if c in (8, 12, 16):
                assert (key.commit(sc) == before).all(), c                      # host scalars
    finally:
        gpu_lib.tune(_lib.TUNE_TABLE_WIDTH, -1)
    assert (key.commit_device(d, n) == before).all()
    gpu_lib.check(gpu_lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
    assert 8 <= tb.value <= 16
    pa, ca, wa = key.commit_partial_device(0, d, m)
    gpu_lib.check(gpu_lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
    pb, cb, wb = key.commit_partial_device(m, d + m * 32, n - m)
    assert (ca, wa) == (0, 1) == (cb, wb) and tb.value == 16
    assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa) == before).all()
    gpu_lib.free(d); gpu_lib.free(dw); key.close()


@pytest.mark.parametrize("cid,n,c", [(0, 1 << 17, 8), (1, 1 << 15, 9), (0, 1 << 17, 6), (1, 1 << 13, 5), (0, 20000, 4)])
def test_every_bucket_heavy(gpu_lib, cid, n, c):
    """Dense scalars under narrow windows cut EVERY bucket into a heavy run; k_fixup_heavy_a then sizes its sub-jobs by their
    number: 4 096 sub-jobs of 47 partials (4 quads each), 7 424 of 7 (2 quads), 5 504 of ~30 (2 quads), 832 of 31 (16 quads),
    1 536 (8 quads).  Every point against the oracle; the width is the key's own (mira_msm_set_handle_window_bits)."""
    key = cm.CommitmentKey.synthetic(cid, n, seed=600 + c)
    key.set_window_bits(c)
    d = cm.synth_scalars_device(cid, n, seed=610 + c)
    sc = gpu_lib.download(d, (n, 4))
    want = C.msm_pippenger(cid, sc, key.download())
    assert (key.commit_device(d, n) == want).all()
    assert (key.commit(sc) == want).all()                        # host scalars, same width
    gpu_lib.free(d); key.close()


@pytest.mark.parametrize("cid,log_n", [(0, 17), (1, 20)])
def test_glv_split_every_width(gpu_lib, cid, log_n):
    """mira_msm_precompute_ex(handle, MIRA_TABLE_GLV): the endomorphism copy of the key, every scalar split into two signed
    127-bit halves (glv.cuh).  The plain per-window path's point (which the oracle confirms) for a dense and a witness-like
    vector and a vector of the decomposition's edge scalars, planned and under every width 5 .. 16 (ceil(128 / c) windows), a
    prefix, host scalars, a batch of five; chunk partials keep the plain shape; a table set beside it takes precedence."""
    n = 1 << log_n
    r = P.CURVES[cid].r
    key = cm.CommitmentKey.synthetic(cid, n, seed=141)
    d = cm.synth_scalars_device(cid, n, seed=142)
    dw = cm.synth_scalars_device(cid, n, seed=143, kind=1)
    sc = gpu_lib.download(d, (n, 4))
    edge = [0, 1, 2, (1 << 126) - 1, 1 << 126, (1 << 127) - 1, 1 << 127, (1 << 128) + 5, r - 1, r - 2, 1 << 253, (1 << 253) + 12345]
    se = sc.copy()
    se[1000:1000 + 64 * len(edge)] = np.tile(ints_to_mont(edge, r), (64, 1))
    de = gpu_lib.alloc(n * 32); gpu_lib.upload(de, se)
    before, before_w, before_e = key.commit_device(d, n), key.commit_device(dw, n), key.commit_device(de, n)
    if log_n <= 17:
        assert (before == C.commit(cid, key.bases(), sc)).all() and (before_e == C.commit(cid, key.bases(), se)).all()
    m = n // 2 + 1234
    before_m = key.commit_device(d, m)
    before_b = key.commit_batch_device(d, n // 8, 5, stride=n // 8 + 7)
    cc, ww = ctypes.c_int32(), ctypes.c_int32()

    def last_plan():
        gpu_lib.check(gpu_lib.c.mira_msm_last_plan(ctypes.byref(cc), ctypes.byref(ww)))
        return cc.value, ww.value
    key.precompute(_lib.TABLE_GLV)
    try:
        assert (key.commit_device(d, n) == before).all()
        c0, w0 = last_plan()
        # planned widths: both planners are asked and the split is taken where it is estimated ahead (up to ~2^19 pairs)
        assert w0 == (-(-128 // c0) if log_n <= 18 else -(-256 // c0))
        assert (key.commit_device(dw, n) == before_w).all() and (key.commit_device(dw, n) == before_w).all()   # the second one planned by the first one's statistics
        assert (key.commit_device(de, n) == before_e).all()
        assert (key.commit_device(d, m) == before_m).all()
        assert (key.commit(sc) == before).all()                                   # host scalars, in point chunks at 2^20
        for c in range(5, 17):
            gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(c))
            assert (key.commit_device(de, n) == before_e).all(), c
            assert last_plan() == (c, -(-128 // c))
            if c in (8, 13, 16):
                assert (key.commit_device(dw, n) == before_w).all(), c
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
        assert (key.commit_batch_device(d, n // 8, 5, stride=n // 8 + 7) == before_b).all()
        assert last_plan()[1] in (-(-128 // last_plan()[0]), -(-256 // last_plan()[0]))
        for c in (8, 13, 16):
            gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(c))
            assert (key.commit_batch_device(d, n // 8, 5, stride=n // 8 + 7) == before_b).all(), c
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
        pa, ca, wa = key.commit_partial_device(0, d, m)
        pb, cb, wb = key.commit_partial_device(m, d + m * 32, n - m)
        assert (ca, wa) == (cb, wb) and wa == -(-256 // ca)
        assert (cm.combine_partials(cid, np.stack([pa, pb]), ca, wa) == before).all()
        gpu_lib.tune(_lib.TUNE_GLV, 0)
        assert (key.commit_device(d, n) == before).all() and last_plan()[1] == -(-256 // last_plan()[0])
        gpu_lib.tune(_lib.TUNE_GLV, -1)
        key.precompute(13)                                                        # a shared-bucket set beside it serves the commit
        tb = ctypes.c_int32()
        assert (key.commit_device(d, n) == before).all()
        gpu_lib.check(gpu_lib.c.mira_msm_last_table_bits(ctypes.byref(tb)))
        assert tb.value == 13
    finally:
        gpu_lib.check(gpu_lib.c.mira_msm_set_window_bits(0))
        gpu_lib.tune(_lib.TUNE_GLV, -1)
        gpu_lib.free(d); gpu_lib.free(dw); gpu_lib.free(de); key.close()
