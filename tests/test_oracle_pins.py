"""Pins the CPU oracle (oracle/pyref.py and oracle/oracle.c) to every known-answer vector the
reference holds for this path, before anything else trusts it."""
import numpy as np

from helpers import ints_to_mont, load_golden, mont_to_ints, point_to_arr, arr_to_point
from oracle import cref as C
from oracle import pyref as P

KATS = load_golden("ref_kats.json")


def test_fr_modulus_matches_reference_literal():
    # src/digest.rs:103 holds MODULUS - 1 of bn256::Fr in decimal
    assert int(KATS["fr_modulus_minus_one"]["decimal"]) == P.R_MOD - 1


def test_fq_modulus_from_bn_parametrisation_and_g2_generator():
    t = P.BN_T
    assert P.R_MOD == 36 * t**4 + 36 * t**3 + 18 * t**2 + 6 * t + 1
    assert P.P_MOD == 36 * t**4 + 36 * t**3 + 24 * t**2 + 6 * t + 1
    # the G2 generator the reference hard-codes (src/gadgets/ecc2.rs:159-176) lies on the twist
    # y^2 = x^3 + 3/(9+u) over Fq2 = Fq[u]/(u^2+1) -- true only for the right Fq modulus
    p = P.P_MOD
    x = tuple(int(v) for v in KATS["bn254_g2_generator"]["x"])
    y = tuple(int(v) for v in KATS["bn254_g2_generator"]["y"])
    mul2 = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
    d = pow(9 * 9 + 1, -1, p)
    b2 = mul2((3, 0), (9 * d % p, (-d) % p))
    x3 = mul2(mul2(x, x), x)
    assert mul2(y, y) == ((x3[0] + b2[0]) % p, (x3[1] + b2[1]) % p)


def test_fft_simple_input_kat_python():
    k = KATS["fft_simple_input_test"]
    a = list(k["input"])
    P.fft(a, k["log_n"])
    assert a == [int(v) for v in k["output_decimal"]]


def test_fft_simple_input_kat_c():
    k = KATS["fft_simple_input_test"]
    a = ints_to_mont(k["input"], P.R_MOD)
    for threads in (1, 2, 8):      # iterative branch and recursive branch of best_fft
        out = C.fft(a, k["log_n"], threads)
        assert mont_to_ints(out, P.R_MOD) == [int(v) for v in k["output_decimal"]]


def test_fft_random_input_roundtrip():
    # src/fft.rs:265-279 (the reference fills the vector with one repeated element; do both)
    for k in (4, 5, 6, 7, 8):
        rep = np.repeat(C.synth_scalars(0, 1, seed=k), 1 << k, axis=0)
        rnd = C.synth_scalars(0, 1 << k, seed=100 + k)
        for a in (rep, rnd):
            assert (C.ifft(C.fft(a, k), k) == a).all()
            lst = mont_to_ints(a, P.R_MOD)
            b = list(lst); P.fft(b, k); P.ifft(b, k)
            assert b == lst


def test_basic_lagrange_kat():
    # src/polynomial/lagrange.rs:115-127 basic_lagrange_test: L_i(2) over the 4-element cyclic subgroup
    k = KATS["basic_lagrange_test"]
    assert P.pg_lagrange(k["X"], k["log_n"]) == [int(v) for v in k["output_decimal"]]


def test_g1_scalar_mul_kat():
    # src/digest.rs:98-113: into_curve_from_bits(MODULUS-1) == -G1Affine::generator()
    cv = P.BN256
    assert P.ec_mul(P.R_MOD - 1, cv.gen, cv) == (1, P.P_MOD - 2)
    g = C.generator(0)
    assert arr_to_point(g, 0) == (1, 2)
    k = ints_to_mont([P.R_MOD - 1], P.R_MOD)[0]
    assert arr_to_point(C.ec_mul(0, k, g), 0) == (1, P.P_MOD - 2)
    # and through both MSM restatements
    for fn in (C.msm_naive, C.msm_pippenger):
        assert arr_to_point(fn(0, k.reshape(1, 4), g.reshape(1, 8)), 0) == (1, P.P_MOD - 2)


def test_commit_homomorphism_identities():
    # src/plonk/mod.rs:547-557 / src/nifs/vanilla/tests.rs:189,228:
    # Com(W1 + r*W2) == Com(W1) + r*Com(W2), checked on affine points
    for cid in (0, 1):
        cv = P.CURVES[cid]
        n = 64
        ck = C.synth_bases(cid, n, seed=5)
        w1 = mont_to_ints(C.synth_scalars(cid, n, seed=1, kind=1), cv.r)
        w2 = mont_to_ints(C.synth_scalars(cid, n, seed=2), cv.r)
        r = P.synth_scalar(3, cv.r)
        folded = ints_to_mont([(a + r * b) % cv.r for a, b in zip(w1, w2)], cv.r)
        lhs = arr_to_point(C.commit(cid, ck, folded), cid)
        c1 = arr_to_point(C.commit(cid, ck, ints_to_mont(w1, cv.r)), cid)
        c2 = arr_to_point(C.commit(cid, ck, ints_to_mont(w2, cv.r)), cid)
        assert lhs == P.ec_add(c1, P.ec_mul(r, c2, cv), cv)


def test_grumpkin_internal_consistency_unpinned():
    # No reference vector pins Grumpkin: only the cycle property is checked (order of the
    # group y^2 = x^3 - 17 over Fr is the BN256 base modulus).
    cv = P.GRUMPKIN
    assert P.on_curve(cv.gen, cv)
    assert P.ec_mul(P.P_MOD - 1, cv.gen, cv) == P.ec_neg(cv.gen, cv)
    assert pow(P.FR_ZETA, 3, P.R_MOD) == 1 and P.FR_ZETA != 1
