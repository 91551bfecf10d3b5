"""SURVEY.md 8(f) rows N2 (witness / error folding) and N3 (commitment-key cache) -- oracle
cross-checks on CPU and the kernel logic under the test-only emulation."""
import os

import numpy as np
import pytest

from helpers import arr_to_point, ints_to_mont, mont_to_ints, point_to_arr
from mira_amd import commitment as cm
from mira_amd import fold as FD
from oracle import cref as C
from oracle import pyref as P

MODS = {0: P.P_MOD, 1: P.R_MOD}          # field id -> modulus
CURVE_OF_FIELD = {1: 0, 0: 1}            # scalars of bn256 live in Fr (1), of grumpkin in Fq (0)


@pytest.mark.parametrize("field", [0, 1])
def test_oracle_fold_c_vs_python(field):
    mod, cid, n = MODS[field], CURVE_OF_FIELD[field], 257
    w1, w2 = C.synth_scalars(cid, n, seed=1, kind=1), C.synth_scalars(cid, n, seed=2)
    r = C.synth_scalars(cid, 1, seed=3)[0]
    ri = mont_to_ints(r, mod)[0]
    want = P.fold_witness(mont_to_ints(w1, mod), mont_to_ints(w2, mod), ri, mod)
    assert mont_to_ints(C.fold_witness(field, w1, w2, r), mod) == want
    terms = [C.synth_scalars(cid, n, seed=10 + k) for k in range(5)]
    want_e = P.fold_error(mont_to_ints(w1, mod), [mont_to_ints(t, mod) for t in terms], ri, mod)
    assert mont_to_ints(C.fold_error(field, w1, terms, r), mod) == want_e


@pytest.mark.parametrize("field", [0, 1])
def test_emu_fold(emu_lib, field):
    cid, n = CURVE_OF_FIELD[field], 1000
    w1, w2 = C.synth_scalars(cid, n, seed=4, kind=1), C.synth_scalars(cid, n, seed=5)
    r = C.synth_scalars(cid, 1, seed=6)[0]
    assert (FD.fold_witness(field, w1, w2, r, lib=emu_lib) == C.fold_witness(field, w1, w2, r)).all()
    terms = [C.synth_scalars(cid, n, seed=20 + k) for k in range(6)]
    assert (FD.fold_error(field, w1, terms, r, lib=emu_lib) == C.fold_error(field, w1, terms, r)).all()
    assert (FD.fold_error(field, w1, [], r, lib=emu_lib) == w1).all()
    with pytest.raises(AssertionError):
        FD.fold_witness(field, w1, w2[:-1], r, lib=emu_lib)        # zip_eq


def _relaxed_fold_case(lib, field, n_w, n, nterms, in_place):
    """mira_fold_relaxed_witness_device against the oracle's fold_witness / fold_error (src/plonk/mod.rs:1097-1134)."""
    cid = CURVE_OF_FIELD[field]
    w1, w2 = C.synth_scalars(cid, max(n_w, 1), seed=71, kind=1)[:n_w], C.synth_scalars(cid, max(n_w, 1), seed=72)[:n_w]
    e = C.synth_scalars(cid, max(n, 1), seed=73)[:n]
    terms = [C.synth_scalars(cid, max(n, 1), seed=80 + k)[:n] for k in range(nterms)]
    r = C.synth_scalars(cid, 1, seed=74)[0]
    up = lambda a: (lambda d: (lib.upload(d, np.ascontiguousarray(a)), d)[1])(lib.alloc(max(32, a.nbytes)))
    d_w1, d_w2, d_e = up(w1), up(w2), up(e)
    d_t = [up(t) for t in terms]
    d_w_out, d_e_out = lib.alloc(max(32, n_w * 32)), (d_e if in_place else lib.alloc(max(32, n * 32)))
    try:
        FD.fold_relaxed_witness_device(field, d_w_out, d_w1, d_w2, n_w, d_e_out, d_e, d_t, r, n, lib=lib)
        if n_w:
            assert (lib.download(d_w_out, (n_w, 4)) == C.fold_witness(field, w1, w2, r)).all()
        if n:
            assert (lib.download(d_e_out, (n, 4)) == (C.fold_error(field, e, terms, r) if nterms else e)).all()
            if not in_place:
                assert (lib.download(d_e, (n, 4)) == e).all()             # the accumulator's E is left alone
    finally:
        for d in [d_w1, d_w2, d_e, d_w_out] + d_t + ([] if in_place else [d_e_out]):
            lib.free(d)


@pytest.mark.parametrize("field", [0, 1])
def test_emu_fold_relaxed_witness_in_one_submission(emu_lib, field):
    for n_w, n, nterms, in_place in ((1400, 200, 6, False), (700, 100, 5, True), (0, 64, 3, False), (300, 0, 0, False), (100, 50, 0, False), (100, 50, 0, True)):
        _relaxed_fold_case(emu_lib, field, n_w, n, nterms, in_place)


def test_emu_commit_is_homomorphic_over_fold(emu_lib):
    """is_sat_relaxed's check (src/plonk/mod.rs:547-557): the commitment of the folded witness
    equals the folded commitment -- with every piece coming from this library."""
    for cid, field in ((0, 1), (1, 0)):
        n = 64
        bases = C.synth_bases(cid, n, seed=3)
        key = cm.CommitmentKey(cid, bases, lib=emu_lib)
        w1, w2 = C.synth_scalars(cid, n, seed=7, kind=1), C.synth_scalars(cid, n, seed=8)
        r = C.synth_scalars(cid, 1, seed=9)[0]
        folded = FD.fold_witness(field, w1, w2, r, lib=emu_lib)
        lhs = key.commit(folded)
        rhs = FD.g1_mul_add(cid, key.commit(w1), r, key.commit(w2), lib=emu_lib)
        assert (lhs == rhs).all()


def test_emu_g1_mul_add_matches_oracle(emu_lib):
    for cid in (0, 1):
        cv = P.CURVES[cid]
        pts = C.synth_bases(cid, 2, seed=12)
        s = C.synth_scalars(cid, 1, seed=13)[0]
        want = P.ec_add(arr_to_point(pts[0], cid), P.ec_mul(mont_to_ints(s, cv.r)[0], arr_to_point(pts[1], cid), cv), cv)
        assert arr_to_point(FD.g1_mul_add(cid, pts[0], s, pts[1], lib=emu_lib), cid) == want
        zero = np.zeros(8, dtype=np.uint64)
        assert (FD.g1_mul_add(cid, zero, np.zeros(4, dtype=np.uint64), pts[1], lib=emu_lib) == zero).all()


def test_emu_g1_lincomb_matches_oracle_and_mul_add(emu_lib):
    """mira_g1_lincomb (shared doublings) = the chain of g1_mul_add calls it replaces = the oracle's
    MSM of the same terms plus the accumulator; edge cases: identity points, zero and r - 1 scalars,
    a repeated point, no terms."""
    for cid in (0, 1):
        cv = P.CURVES[cid]
        pts = C.synth_bases(cid, 7, seed=14)
        sc = C.synth_scalars(cid, 6, seed=15)
        sc[2] = 0
        sc[3] = ints_to_mont([cv.r - 1], cv.r)[0]
        terms = pts[1:].copy()
        terms[4] = 0                                           # an identity term
        terms[5] = terms[0]                                    # a repeated point
        got = FD.g1_lincomb(cid, pts[0], sc, terms, lib=emu_lib)
        chain = pts[0]
        for s_, t in zip(sc, terms):
            chain = FD.g1_mul_add(cid, chain, s_, t, lib=emu_lib)
        assert (got == chain).all()
        assert (got == C.ec_add(cid, pts[0], C.msm_naive(cid, sc, terms))).all()
        zero = np.zeros(8, dtype=np.uint64)
        assert (FD.g1_lincomb(cid, zero, sc[:0], terms[:0], lib=emu_lib) == zero).all()
        assert (FD.g1_lincomb(cid, pts[0], sc[:0], terms[:0], lib=emu_lib) == pts[0]).all()


def test_emu_g1_mul_add_naf_edge_scalars(emu_lib):
    """The width-5 NAF behind every host scalar multiplication: scalars around its digit boundaries (15 / 16 / 17, 31 / 32 / 33),
    all-ones low bits (a carry that runs through the limbs), r - 1 and r - 16, against the oracle's double-and-add."""
    for cid in (0, 1):
        cv = P.CURVES[cid]
        pts = C.synth_bases(cid, 2, seed=21)
        ks = [1, 2, 3, 15, 16, 17, 31, 32, 33, (1 << 64) - 1, (1 << 64) + 15, (1 << 128) - 1, (1 << 192) - 17, (1 << 253) + (1 << 64) - 1, cv.r - 1, cv.r - 16, cv.r - 17]
        for k, s in zip(ks, ints_to_mont(ks, cv.r)):
            want = P.ec_add(arr_to_point(pts[0], cid), P.ec_mul(k, arr_to_point(pts[1], cid), cv), cv)
            assert arr_to_point(FD.g1_mul_add(cid, pts[0], s, pts[1], lib=emu_lib), cid) == want, k


def _instance_fold_case(cid, lib, nw, count, seed):
    cv = P.CURVES[cid]
    pts = C.synth_bases(cid, 2 * nw + count + 1, seed=seed)
    r = C.synth_scalars(cid, 1, seed=seed + 1)[0]
    w1, w2, e, t = pts[:nw], pts[nw:2 * nw], pts[2 * nw], pts[2 * nw + 1:]
    w_out, e_out = FD.fold_instance_commitments(cid, w1, w2, r, e, t, lib=lib)
    r_int = mont_to_ints(r, cv.r)[0]
    powers = ints_to_mont([pow(r_int, k + 1, cv.r) for k in range(count)], cv.r) if count else np.zeros((0, 4), dtype=np.uint64)
    for i in range(nw):
        assert (w_out[i] == C.ec_add(cid, w1[i], C.msm_naive(cid, r[None, :], w2[i:i + 1]))).all()
    want_e = C.ec_add(cid, e, C.msm_naive(cid, powers, t)) if count else e
    assert (e_out == want_e).all()
    return w_out, e_out


def test_emu_fold_instance_commitments_matches_oracle(emu_lib):
    """mira_g1_fold_commitments = RelaxedPlonkInstance::fold's commitments (src/plonk/mod.rs:986-999, 1049-1053):
    W1_i + r W2_i and E + sum_k r^(k+1) T_k against the oracle; no W commitments, no cross terms."""
    for cid in (0, 1):
        for nw, count in ((1, 5), (2, 6), (0, 3), (3, 0), (0, 0)):
            _instance_fold_case(cid, emu_lib, nw, count, seed=30 + nw * 7 + count)


def test_product_host_fold_runs_on_its_threads():
    """The same calls through libmira_gpu.so: these entry points are host arithmetic (no device call), and in the product
    build their terms are dealt to the library's resident threads -- the points must not depend on how."""
    from mira_amd import _lib
    lib = _lib.load()
    for cid in (0, 1):
        for nw, count in ((1, 6), (2, 5), (1, 13), (0, 1), (4, 0)):
            _instance_fold_case(cid, lib, nw, count, seed=50 + nw * 5 + count)
        pts = C.synth_bases(cid, 10, seed=61)
        sc = C.synth_scalars(cid, 9, seed=62)
        assert (FD.g1_lincomb(cid, pts[0], sc, pts[1:], lib=lib) == C.ec_add(cid, pts[0], C.msm_naive(cid, sc, pts[1:]))).all()


def test_emu_key_cache_roundtrip(emu_lib, tmp_path):
    """src/commitment.rs:96-167 and its test `consistency` (:178-194): save, load, compare."""
    cid, k = 0, 6
    key = cm.CommitmentKey.synthetic(cid, 1 << k, lib=emu_lib)
    original = key.bases()
    assert (key.download() == original).all()               # export inverts the resident conversion
    path = tmp_path / "my-temporary-note.txt"
    key.save_to_file(path)
    assert os.path.getsize(path) == (1 << k) * 64
    assert open(path, "rb").read() == original.tobytes()     # raw dump of the slice
    loaded = cm.CommitmentKey.load_from_file(cid, path, k, lib=emu_lib)
    assert (loaded.download() == original).all()
    with pytest.raises(IOError):
        cm.CommitmentKey.load_from_file(cid, path, k + 1, lib=emu_lib)     # read_exact fails
    # load_or_setup_cache: creates {folder}/{label}/{k}.bin, reloads and validates it
    k1 = cm.CommitmentKey.load_or_setup_cache(cid, str(tmp_path / "cache"), "bn256", k, lib=emu_lib)
    p = tmp_path / "cache" / "bn256" / f"{k}.bin"
    assert p.exists()
    k2 = cm.CommitmentKey.load_or_setup_cache(cid, str(tmp_path / "cache"), "bn256", k, lib=emu_lib)
    assert (k1.download() == k2.download()).all()
    raw = bytearray(p.read_bytes()); raw[70] ^= 1; p.write_bytes(bytes(raw))
    with pytest.raises(IOError, match="Wrong file in cache, some ptr out of curve"):
        cm.CommitmentKey.load_or_setup_cache(cid, str(tmp_path / "cache"), "bn256", k, lib=emu_lib)
