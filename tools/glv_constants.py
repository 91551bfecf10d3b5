"""Development: the GLV constants of the two curves (mira_amd/csrc/glv.cuh, glv_consts.h), derived and checked against the
oracle's curve arithmetic.  Both curves are y^2 = x^3 + b: phi(x, y) = (beta x, y) is an endomorphism and acts on the
prime-order group as multiplication by a cube root of unity lambda of the scalar field.  Prints, per curve: beta, lambda (the
pair with phi(G) = lambda G), a reduced basis (a1, b1), (a2, b2) of the lattice {(x, y): x + y lambda = 0 mod r}, the rounding
constants g_i = round(2^256 b_i / r) of the division-free decomposition, and the largest |k1|, |k2| seen over the test scalars.
usage: python tools/glv_constants.py [--cpp]      (--cpp: the constants as mira_amd/csrc/glv_consts.h holds them)"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref as P


def cube_roots_of_unity(mod):
    for g in range(2, 50):
        w = pow(g, (mod - 1) // 3, mod)
        if w != 1:
            return w, w * w % mod
    raise ValueError("no generator found")


def lattice_basis(r, lam):
    """Extended Euclid on (r, lambda) stopped around sqrt(r) (Gallant-Lambert-Vanstone, section 4)."""
    rows = [(r, 0), (lam, 1)]                       # (remainder, t): remainder = s r + t lambda
    while rows[-1][0] * rows[-1][0] >= r:
        (r0, t0), (r1, t1) = rows[-2], rows[-1]
        q = r0 // r1
        rows.append((r0 - q * r1, t0 - q * t1))
    (r0, t0), (r1, t1) = rows[-2], rows[-1]
    q = r0 // r1
    r2, t2 = r0 - q * r1, t0 - q * t1
    v1 = (r1, -t1)
    v2 = (r0, -t0) if r0 * r0 + t0 * t0 <= r2 * r2 + t2 * t2 else (r2, -t2)
    return v1, v2


def decompose(k, r, v1, v2, g1, g2):
    (a1, b1), (a2, b2) = v1, v2
    c1 = (k * g1 + (1 << 255)) >> 256               # round(k b2 / r), the division replaced by a 256-bit shift
    c2 = (k * g2 + (1 << 255)) >> 256               # round(-k b1 / r)
    k1 = k - c1 * a1 - c2 * a2
    k2 = -c1 * b1 - c2 * b2
    return k1, k2


for cid, cv in P.CURVES.items():
    r, p = cv.r, cv.p
    G = P.synth_base(0, cv)
    betas, lams = cube_roots_of_unity(p), cube_roots_of_unity(r)
    pair = [(b, l) for b in betas for l in lams if (b * G[0] % p, G[1]) == P.ec_mul(l, G, cv)]
    assert len(pair) == 2                            # (beta, lambda) and (beta^2, lambda^2)
    beta, lam = min(pair, key=lambda x: x[1])
    assert (lam * lam + lam + 1) % r == 0 and pow(beta, 3, p) == 1
    v1, v2 = lattice_basis(r, lam)
    for (a, b) in (v1, v2):
        assert (a + b * lam) % r == 0
    det = v1[0] * v2[1] - v1[1] * v2[0]
    if det < 0:
        v2 = (-v2[0], -v2[1]); det = -det
    assert det == r
    g1 = ((v2[1] << 256) + r // 2) // r if v2[1] >= 0 else -((((-v2[1]) << 256) + r // 2) // r)
    g2 = ((-v1[1] << 256) + r // 2) // r if -v1[1] >= 0 else -(((v1[1] << 256) + r // 2) // r)
    rng = random.Random(7 + cid)
    worst = 0
    ks = [0, 1, 2, r - 1, r - 2, lam, r - lam, (1 << 128) - 1, 1 << 128, 1 << 253] + [rng.randrange(r) for _ in range(2000)]
    for k in ks:
        k1, k2 = decompose(k, r, v1, v2, g1, g2)
        assert (k1 + k2 * lam - k) % r == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    for k in ks[:14]:                                # the point identity, on the oracle's arithmetic
        Q = P.synth_base(3, cv)
        k1, k2 = decompose(k, r, v1, v2, g1, g2)
        phiQ = (beta * Q[0] % p, Q[1])
        part = lambda s, pt: P.ec_mul(abs(s), pt if s >= 0 else P.ec_neg(pt, cv), cv)
        assert P.ec_add(part(k1, Q), part(k2, phiQ), cv) == P.ec_mul(k, Q, cv)
    if "--cpp" in sys.argv:                          # the specialisation of Glv<> in mira_amd/csrc/glv_consts.h
        limbs = lambda v, n: ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))
        assert v1[0] > 0 and v1[1] < 0 and v2[0] > 0 and v2[1] > 0 and g1 > 0 and g2 > 0
        print("template <> struct Glv<%s> {                                  // curve %d: scalars in %s, lambda = 0x%x" % ("FrP" if cid == 0 else "FqP", cid, "bn256::Fr" if cid == 0 else "bn256::Fq", lam))
        print("    static constexpr uint32_t G1[3] = {%s}, G2[5] = {%s};" % (limbs(g1, 3), limbs(g2, 5)))
        print("    static constexpr uint32_t A1[5] = {%s}, A2[5] = {%s};" % (limbs(v1[0], 5), limbs(v2[0], 5)))
        print("    static constexpr uint32_t NB1[5] = {%s}, B2[5] = {%s};   // -b1, b2" % (limbs(-v1[1], 5), limbs(v2[1], 5)))
        print("    static constexpr uint64_t BETA[4] = {%s};                // beta, a plain integer of the base field" % ", ".join("0x%016xull" % ((beta >> (64 * i)) & (2 ** 64 - 1)) for i in range(4)))
        print("};")
        continue
    print("curve %d (%s)" % (cid, "bn256 G1" if cid == 0 else "grumpkin"))
    print("  beta   = 0x%064x" % beta)
    print("  lambda = 0x%064x" % lam)
    print("  v1 = (%d, %d)\n  v2 = (%d, %d)" % (v1 + v2))
    print("  g1 = %s0x%x\n  g2 = %s0x%x" % ("-" if g1 < 0 else "", abs(g1), "-" if g2 < 0 else "", abs(g2)))
    print("  |k1|, |k2| < 2^%d over %d scalars; k = k1 + k2 lambda and k Q = k1 Q + k2 phi(Q) hold" % (worst, len(ks)))
