"""TEST INFRASTRUCTURE ONLY -- Python-integer restatement of the reference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (mira_amd/) never does: it fails loudly when the HIP library is
missing.

What is restated, with the reference lines each function follows
(paths relative to /root/reference):

* radix-2 NTT                 src/fft.rs:12-27 (omega derivation), :51-115 (best_fft),
                              :118-155 (recursive butterflies), :160-174 (fft/ifft),
                              :178-226 (coset variants)
* CommitmentKey::commit       src/commitment.rs:78-87 (prefix semantics, length check,
                              affine output).  The MSM arithmetic itself lives in the
                              un-vendored git dependency halo2_proofs (Cargo.toml:60-62,
                              branch joshbeal/dev-mira, no lock file) on top of
                              halo2curves; its *result* is a unique group element, so it is
                              restated here as plain double-and-add.
* concatenate_with_padding    src/util.rs:189-193
* cross-term evaluation       src/polynomial/expression.rs:112-120 evaluated directly (eval_expression),
                              src/plonk/eval.rs:152-206 (plonk_advice_location)
* cross-term expressions      src/plonk/util.rs:97-117 (compress_expression), src/polynomial/expression.rs:233-260,
                              356-430 (fold_transform, homogeneous), src/polynomial/grouped_poly.rs:88-285 (GroupedPoly),
                              src/main_gate.rs:543-589 (the MainGate<T> gate) -- PINNED by the reference's Display tests
* ProtoGalaxy polynomials     src/nifs/protogalaxy/poly/mod.rs:66-179, 218-303, 339-382,
                              folded_trace.rs, src/polynomial/lagrange.rs (pg_*)

Pinning (tests/test_oracle_pins.py):
* NTT: the 8-point known-answer vector of src/fft.rs:240-257, and the ifft(fft(x)) == x
  property of src/fft.rs:265-279.
* Fr modulus: src/digest.rs:101-105 holds r-1 in decimal.
* Fq modulus: derived from r through the BN parametrisation (t = 4965661367192848881);
  the BN254 G2 generator held in src/gadgets/ecc2.rs:156-180 lies on the sextic twist
  y^2 = x^3 + 3/(9+u) over Fq2 only for this p.
* BN256 G1 (b = 3, generator (1,2), scalar mul): src/digest.rs:98-113 asserts
  (r-1)*G == -G.
* MSM: no known-answer vector exists in the reference.  Pinned by the homomorphic
  identities the reference's folding tests assert on affine points
  (src/nifs/vanilla/tests.rs:189,228 ; src/plonk/mod.rs:547-557):
  Com(W1 + r*W2) == Com(W1) + r*Com(W2).
* Grumpkin (y^2 = x^3 - 17 over Fr, order p), Fr::ZETA, CommitmentKey::setup output:
  PARITY UNPINNED by any reference vector (checked here only for internal consistency:
  group order, zeta^3 = 1).
"""

from __future__ import annotations

# ----------------------------------------------------------------------------- fields
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # bn256::Fr
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # bn256::Fq
BN_T = 4965661367192848881
MONT_R = 1 << 256

FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)
FR_ROOT_OF_UNITY_INV = pow(FR_ROOT_OF_UNITY, R_MOD - 2, R_MOD)
FR_TWO_INV = pow(2, R_MOD - 2, R_MOD)
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23

CURVE_BN256 = 0     # y^2 = x^3 + 3 over Fq, scalars in Fr
CURVE_GRUMPKIN = 1  # y^2 = x^3 - 17 over Fr, scalars in Fq


class Curve:
    def __init__(self, cid, base_mod, scalar_mod, b, gen):
        self.id, self.p, self.r, self.b, self.gen = cid, base_mod, scalar_mod, b % base_mod, gen


BN256 = Curve(CURVE_BN256, P_MOD, R_MOD, 3, (1, 2))
GRUMPKIN = Curve(CURVE_GRUMPKIN, R_MOD, P_MOD, -17,
                 (1, 17631683881184975370165255887551781615748388533673675138860))
CURVES = {CURVE_BN256: BN256, CURVE_GRUMPKIN: GRUMPKIN}


def to_mont(x, mod):
    return (x * MONT_R) % mod


def from_mont(x, mod):
    return (x * pow(MONT_R, -1, mod)) % mod


def limbs4(x):
    """256-bit int -> 4 little-endian u64 limbs (the in-memory layout of halo2curves fields)."""
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs4(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ----------------------------------------------------------------------------- curve
def on_curve(P, cv):
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - cv.b) % cv.p == 0


def ec_neg(P, cv):
    return None if P is None else (P[0], (-P[1]) % cv.p)


def ec_add(P, Q, cv):
    if P is None:
        return Q
    if Q is None:
        return P
    p = cv.p
    if P[0] == Q[0]:
        if (P[1] + Q[1]) % p == 0:
            return None
        lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p) % p
    else:
        lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    x = (lam * lam - P[0] - Q[0]) % p
    return (x, (lam * (P[0] - x) - P[1]) % p)


def ec_mul(k, P, cv):
    k %= cv.r
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, P, cv)
        P = ec_add(P, P, cv)
        k >>= 1
    return acc


def msm_naive(scalars, bases, cv):
    """sum_i scalars[i] * bases[i]; scalars canonical ints, bases affine tuples or None."""
    acc = None
    for s, B in zip(scalars, bases):
        acc = ec_add(acc, ec_mul(s, B, cv), cv)
    return acc


class TooLongInput(Exception):
    """src/commitment.rs:21-24"""

    def __init__(self, input_len, limit):
        super().__init__(f"Can't commit too long input: input len: {input_len}, but limit is {limit}")
        self.input_len, self.limit = input_len, limit


def commit(ck, v, cv):
    """src/commitment.rs:78-87: MSM over the PREFIX of the key, affine result."""
    if len(ck) >= len(v):
        return msm_naive(v, ck[:len(v)], cv)
    raise TooLongInput(len(v), len(ck))


def concatenate_with_padding(vs, pad_size):
    """src/util.rs:189-193: each column zero-padded to pad_size, columns concatenated."""
    out = []
    for v in vs:
        out.extend(v)
        out.extend([0] * max(0, pad_size - len(v)))
    return out


# ----------------------------------------------------------------------------- witness folding
def fold_witness(w1, w2, r, mod):
    """src/plonk/mod.rs:1099-1110: w1 + r * w2, element-wise"""
    assert len(w1) == len(w2)
    return [(a + r * b) % mod for a, b in zip(w1, w2)]


def fold_error(e, cross_terms, r, mod):
    """src/plonk/mod.rs:1118-1131: e_i + sum_k r^(k+1) * T_k[i]"""
    out = []
    for i, ei in enumerate(e):
        acc, pw = ei, r
        for t in cross_terms:
            acc = (acc + pw * t[i]) % mod
            pw = pw * r % mod
        out.append(acc)
    return out


# ---------------------------------------------------------------- cross-term evaluation
# Direct evaluation of an Expression tree (reference src/polynomial/expression.rs:112-120) with
# Python integers -- independent of the calculation graph the product builds from the same tree
# (src/polynomial/graph_evaluator.rs:261-352), so it checks graph construction and the device VM
# together.  Trees are nested tuples:
#   ("const", v) ("poly", index, rotation) ("chal", index) ("neg", a) ("sum", a, b)
#   ("prod", a, b) ("scaled", a, f)            v, f: plain integers mod `mod`
# getter: dict(selectors=[[bool]], fixed=[[int]], advice=[[int]], challenges=[int]); column index
# space = selectors, then fixed, then advice (eval_column_var, src/plonk/eval.rs:57-69).
def plonk_advice_location(num_advice, num_lookup, len_w1s, len_w2s, index):
    """PlonkEvalDomain::eval_advice_var's index arithmetic (src/plonk/eval.rs:152-206):
    advice index -> (is_first_instance, witness vector i, column j inside it), or None where the
    reference returns Error::InvalidWitnessIndex for the witness count."""
    max_width = num_advice + num_lookup * 5
    is_first = index < max_width
    if not is_first:
        index -= max_width
    num_witness = len_w1s if is_first else len_w2s
    if index < num_advice:
        return (is_first, 0, index)
    lookup_index, lookup_sub = (index - num_advice) // 5, (index - num_advice) % 5
    first_round = lookup_sub < 3
    if not first_round:
        lookup_sub -= 3
    if num_witness == 2:
        return (is_first, 0, num_advice + lookup_index * 3 + lookup_sub) if first_round else (is_first, 1, lookup_index * 2 + lookup_sub)
    if num_witness == 3:
        return (is_first, 1, lookup_index * 3 + lookup_sub) if first_round else (is_first, 2, lookup_index * 2 + lookup_sub)
    return None


def eval_expression(expr, getter, row, num_rows, mod):
    kind = expr[0]
    if kind == "const":
        return expr[1] % mod
    if kind == "poly":
        index, rot = expr[1], expr[2]
        r = (row + rot) % num_rows                       # rem_euclid, graph_evaluator.rs:51-53
        sel, fix, adv = getter["selectors"], getter["fixed"], getter["advice"]
        if index < len(sel):
            return 1 if sel[index][r] else 0
        index -= len(sel)
        if index < len(fix):
            return fix[index][r] % mod
        index -= len(fix)
        if index >= len(adv):
            raise IndexError("column variable index out of boundary")
        return adv[index][r] % mod
    if kind == "chal":
        if expr[1] >= len(getter["challenges"]):
            raise IndexError("challenge index out of boundary")
        return getter["challenges"][expr[1]] % mod
    if kind == "neg":
        return (-eval_expression(expr[1], getter, row, num_rows, mod)) % mod
    if kind == "sum":
        return (eval_expression(expr[1], getter, row, num_rows, mod) + eval_expression(expr[2], getter, row, num_rows, mod)) % mod
    if kind == "prod":
        return eval_expression(expr[1], getter, row, num_rows, mod) * eval_expression(expr[2], getter, row, num_rows, mod) % mod
    if kind == "scaled":
        return eval_expression(expr[1], getter, row, num_rows, mod) * expr[2] % mod
    raise ValueError(kind)


# ---------------------------------------------------------------- from the gates to the cross-term expressions
# The symbolic pipeline behind commit_cross_terms' graphs (src/nifs/vanilla/mod.rs:100-104:
# S.custom_gates_lookup_compressed.grouped().iter_from_first()), restated on the nested tuples above -- an
# implementation independent of the product-side mirror (mira_amd/expression.py, grouped_poly.py, main_gate.py),
# which the tests compare with it node for node.  Pinned by the reference's own `Display` tests
# (src/polynomial/expression.rs:528-606, src/polynomial/grouped_poly.rs:287-461; tests/golden/ref_kats.json).
# ctx = dict(num_selectors, num_fixed, num_advice, num_challenges, num_lookups)   (QueryIndexContext, expression.rs:38-67)
def _e_sum(a, b): return ("sum", a, b)
def _e_prod(a, b): return ("prod", a, b)
def _e_neg(a): return ("neg", a)


def expr_to_string(e):
    """`impl Display for Expression` = visualize, src/polynomial/expression.rs:262-301; constants through
    trim_leading_zeros (src/util.rs:160-164: zero prints as `0x`)."""
    hx = lambda v: "0x" + format(v, "x").lstrip("0")
    k = e[0]
    if k == "const":
        return hx(e[1])
    if k == "poly":
        return f"Z_{e[1]}" + ("" if e[2] == 0 else f"[{e[2]}]" if e[2] < 0 else f"[+{e[2]}]")
    if k == "chal":
        return f"r_{e[1]}"
    if k == "neg":
        return "-" + expr_to_string(e[1])
    if k == "sum":
        return expr_to_string(e[1]) + (" - " + expr_to_string(e[2][1]) if e[2][0] == "neg" else " + " + expr_to_string(e[2]))
    if k == "prod":
        side = lambda x: "(" + expr_to_string(x) + ")" if x[0] == "sum" else expr_to_string(x)
        return side(e[1]) + " * " + side(e[2])
    if k == "scaled":
        return '"' + hx(e[2]) + '" * ' + expr_to_string(e[1])
    raise ValueError(k)


def expr_challenges(e):
    """the distinct challenge indices of an expression (num_challenges = their count, expression.rs:160-184)"""
    if e[0] == "chal":
        return {e[1]}
    if e[0] in ("const", "poly"):
        return set()
    return set().union(*[expr_challenges(x) for x in e[1:] if isinstance(x, tuple)])


def query_is_folded(index, ctx):
    """Query::subtype is Advice or Lookup (expression.rs:83-100): the variables that carry degree and get folded"""
    lo = ctx["num_selectors"] + ctx["num_fixed"]
    assert index < lo + ctx["num_advice"] + 5 * ctx["num_lookups"], "unknown index"
    return index >= lo


def challenge_in_degree(index, degree):
    """expression.rs:501-513"""
    r = ("chal", index)
    for _ in range(2, degree + 1):
        r = _e_prod(r, ("chal", index))
    return r


def expr_homogeneous(e, ctx):
    """Expression::homogeneous, expression.rs:356-430 -> (expression, degree)"""
    u = ctx["num_challenges"]
    k = e[0]
    if k == "const":
        return e, 0
    if k == "poly":
        return e, 1 if query_is_folded(e[1], ctx) else 0
    if k == "chal":
        return e, 1
    if k == "neg":
        x, d = expr_homogeneous(e[1], ctx)
        return _e_neg(x), d
    if k == "sum":
        (l, ld), (r, rd) = expr_homogeneous(e[1], ctx), expr_homogeneous(e[2], ctx)
        if ld > rd:
            return _e_sum(l, _e_prod(r, challenge_in_degree(u, ld - rd))), ld
        if ld < rd:
            return _e_sum(_e_prod(l, challenge_in_degree(u, rd - ld)), r), rd
        return _e_sum(l, r), ld
    if k == "prod":
        (l, ld), (r, rd) = expr_homogeneous(e[1], ctx), expr_homogeneous(e[2], ctx)
        return _e_prod(l, r), ld + rd
    if k == "scaled":
        x, d = expr_homogeneous(e[1], ctx)
        return ("scaled", x, e[2]), d
    raise ValueError(k)


def expr_fold_transform(e, mm, nn):
    """Expression::fold_transform, expression.rs:233-260: x_i -> x_i + r y_i for the columns from mm on and for
    every challenge, r = Challenge(2 * num_challenges)"""
    nc = len(expr_challenges(e))
    r = ("chal", 2 * nc)

    def go(x):
        k = x[0]
        if k == "const":
            return x
        if k == "poly":
            return x if x[1] < mm else _e_sum(x, _e_prod(r, ("poly", x[1] + nn, x[2])))
        if k == "chal":
            return _e_sum(x, _e_prod(r, ("chal", x[1] + nc)))
        if k == "neg":
            return _e_neg(go(x[1]))
        if k == "sum":
            return _e_sum(go(x[1]), go(x[2]))
        if k == "prod":
            return _e_prod(go(x[1]), go(x[2]))
        return ("scaled", go(x[1]), x[2])
    return go(e)


# GroupedPoly (src/polynomial/grouped_poly.rs): a list, entry k = coefficient of X^k or None
def grouped_from(pairs):
    """`impl From<(degree, expr) pairs>`, grouped_poly.rs:46-62"""
    t = []
    for d, x in (pairs.items() if isinstance(pairs, dict) else pairs):
        t.extend([None] * (d + 1 - len(t)))
        t[d] = x
    return t


def _grouped_zip(a, b, f):                       # impl_poly_ops!, grouped_poly.rs:171-201
    out = []
    for i in range(max(len(a), len(b))):
        l, r = (a[i] if i < len(a) else None), (b[i] if i < len(b) else None)
        out.append(_e_sum(l, f(r)) if l is not None and r is not None else f(r) if r is not None else l)
    return out


def grouped_add(a, b): return _grouped_zip(a, b, lambda x: x)
def grouped_sub(a, b): return _grouped_zip(a, b, _e_neg)
def grouped_neg(a): return [None if x is None else _e_neg(x) for x in a]                        # :272-285
def grouped_scale(a, k): return [None if x is None else _e_prod(("const", k), x) for x in a]    # Mul<&F>, :203-219


def grouped_mul(a, b):
    """grouped_poly.rs:221-270: the longer operand (the right one on a tie) outside, both from the top degree down"""
    lhs, rhs = (b, a) if len(a) <= len(b) else (a, b)
    res = []
    for ld in range(len(lhs) - 1, -1, -1):
        if lhs[ld] is None:
            continue
        for rd in range(len(rhs) - 1, -1, -1):
            if rhs[rd] is None:
                continue
            d, x = ld + rd, _e_prod(lhs[ld], rhs[rd])
            res.extend([None] * (d + 1 - len(res)))
            res[d] = x if res[d] is None else _e_sum(res[d], x)
    return res


def grouped_new(e, ctx):
    """GroupedPoly::new, grouped_poly.rs:88-140"""
    k = e[0]
    if k == "const":
        return [e]
    if k == "poly":
        shift = ctx["num_advice"] + 5 * ctx["num_lookups"]               # num_fold_vars: shift_advice_index == shift_lookup_index
        return [e, ("poly", e[1] + shift, e[2])] if query_is_folded(e[1], ctx) else [e]
    if k == "chal":
        return [e, ("chal", e[1] + ctx["num_challenges"])]
    if k == "neg":
        return grouped_neg(grouped_new(e[1], ctx))
    if k == "sum":
        return grouped_add(grouped_new(e[1], ctx), grouped_new(e[2], ctx))
    if k == "prod":
        return grouped_mul(grouped_new(e[1], ctx), grouped_new(e[2], ctx))
    if k == "scaled":
        return grouped_scale(grouped_new(e[1], ctx), e[2])
    raise ValueError(k)


def compress_expression(exprs, challenge_index):
    """src/plonk/util.rs:97-117"""
    if len(exprs) > 1:
        acc = ("const", 0)
        for x in exprs:
            acc = _e_sum(x, _e_prod(acc, ("chal", challenge_index)))
        return acc
    return exprs[0] if exprs else ("const", 0)


def compressed_gates(exprs, ctx):
    """CompressedGates::new, src/plonk/mod.rs:92-122 -> dict(compressed, homogeneous, degree, grouped, ctx);
    ctx is a copy updated as the reference updates it"""
    ctx = dict(ctx)
    compressed = compress_expression(exprs, ctx["num_challenges"])
    ctx["num_challenges"] = len(expr_challenges(compressed))
    homogeneous, degree = expr_homogeneous(compressed, ctx)
    ctx["num_challenges"] = len(expr_challenges(homogeneous))
    return dict(compressed=compressed, homogeneous=homogeneous, degree=degree, grouped=grouped_new(homogeneous, ctx), ctx=ctx)


def main_gate_polynomial(T, num_selectors, num_fixed, first_fixed, first_advice):
    """The gate MainGate::<F, T>::configure creates (src/main_gate.rs:543-589) after Expression::from_halo2_expr
    (src/polynomial/expression.rs:303-341): columns in allocation order -- advice state[T], input, out; fixed
    q_1[T], q_5[T], q_m[2], q_i, q_o, rc.
        init = q_m[0] s0 s1 + q_i input + rc + q_o out;  T >= 4: init = q_m[1] s2 s3 + init
        fold(init, + (q_1[i] s_i + q_5[i] (s_i^2 s_i^2) s_i))"""
    fx = lambda c: ("poly", num_selectors + first_fixed + c, 0)
    ad = lambda c: ("poly", num_selectors + num_fixed + first_advice + c, 0)
    st = [ad(i) for i in range(T)]
    q_1, q_5 = [fx(i) for i in range(T)], [fx(T + i) for i in range(T)]
    q_m, q_i, q_o, rc = [fx(2 * T), fx(2 * T + 1)], fx(2 * T + 2), fx(2 * T + 3), fx(2 * T + 4)
    acc = _e_sum(_e_sum(_e_sum(_e_prod(_e_prod(q_m[0], st[0]), st[1]), _e_prod(q_i, ad(T))), rc), _e_prod(q_o, ad(T + 1)))
    if T >= 4:
        acc = _e_sum(_e_prod(_e_prod(q_m[1], st[2]), st[3]), acc)
    for i in range(T):
        v2 = _e_prod(st[i], st[i])
        acc = _e_sum(acc, _e_sum(_e_prod(q_1[i], st[i]), _e_prod(q_5[i], _e_prod(_e_prod(v2, v2), st[i]))))
    return acc


# ----------------------------------------------------------------------------- NTT
def get_omega_or_inv(k, is_inverse):
    """src/fft.rs:12-23"""
    assert k <= FR_S, f"k={k} should no larger than F::S={FR_S}"
    w = FR_ROOT_OF_UNITY_INV if is_inverse else FR_ROOT_OF_UNITY
    for _ in range(k, FR_S):
        w = w * w % R_MOD
    return w


def get_ifft_divisor(k):
    """src/fft.rs:25-27"""
    return pow(FR_TWO_INV, k, R_MOD)


def _bitreverse(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2) if bits else 0


def best_fft(a, omega, log_n):
    """src/fft.rs:51-115 (iterative branch :83-111; the recursive branch :118-155 computes
    the same butterflies in another order).  In place on a list of canonical ints."""
    n = len(a)
    assert n == 1 << log_n
    q = R_MOD
    for k in range(n):
        rk = _bitreverse(k, log_n)
        if k < rk:
            a[rk], a[k] = a[k], a[rk]
    twiddles = [1] * (n // 2)
    for i in range(1, n // 2):
        twiddles[i] = twiddles[i - 1] * omega % q
    chunk, twiddle_chunk = 2, n // 2
    for _ in range(log_n):
        half = chunk // 2
        for base in range(0, n, chunk):
            for i in range(half):
                t = a[base + half + i] * twiddles[i * twiddle_chunk] % q
                u = a[base + i]
                a[base + i] = (u + t) % q
                a[base + half + i] = (u - t) % q
        chunk *= 2
        twiddle_chunk //= 2


def fft(a, log_n):
    """src/fft.rs:160-162"""
    best_fft(a, get_omega_or_inv(log_n, False), log_n)


def ifft(a, log_n):
    """src/fft.rs:165-174"""
    best_fft(a, get_omega_or_inv(log_n, True), log_n)
    d = get_ifft_divisor(log_n)
    for i in range(len(a)):
        a[i] = a[i] * d % R_MOD


def distribute_powers_zeta(a, into_coset):
    """src/fft.rs:205-226"""
    z, zi = FR_ZETA, FR_ZETA * FR_ZETA % R_MOD
    powers = [z, zi] if into_coset else [zi, z]
    for idx in range(len(a)):
        i = idx % 3
        if i:
            a[idx] = a[idx] * powers[i - 1] % R_MOD


def coset_fft(a):
    """src/fft.rs:178-185"""
    log_n = len(a).bit_length() - 1
    assert len(a) == 1 << log_n
    distribute_powers_zeta(a, True)
    fft(a, log_n)


def coset_ifft(a):
    """src/fft.rs:189-196"""
    log_n = len(a).bit_length() - 1
    assert len(a) == 1 << log_n
    ifft(a, log_n)
    distribute_powers_zeta(a, False)
    return list(a)


# ----------------------------------------------------------------------------- synthetic inputs
# One splitmix64 stream per index i, seeded seed + i * STREAM_MUL.  The C oracle
# (oracle_synth_*) and the product's GPU generators (mira_synth_*) use the same definition,
# so inputs can be produced independently on either side and compared bit for bit.
MASK64 = (1 << 64) - 1
STREAM_MUL = 0xD6E8FEB86659FD93
SEED_SCALARS = 0x4D495241
SEED_BASES = 0x42415345


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)


def synth_scalar(i, mod, seed=SEED_SCALARS, kind="uniform"):
    """SURVEY.md 8(d).  kind 'witness': 70 % zero, 20 % < 2^32, 10 % uniform.  Canonical int."""
    g = SplitMix64(seed + i * STREAM_MUL)
    v = sum(g.next() << (64 * k) for k in range(4)) % mod
    if kind == "witness":
        sel = g.next() % 10
        if sel < 7:
            v = 0
        elif sel < 9:
            v &= 0xFFFFFFFF
    return v


def synth_scalars(n, mod, seed=SEED_SCALARS, kind="uniform"):
    return [synth_scalar(i, mod, seed, kind) for i in range(n)]


def synth_base(i, cv, seed=SEED_BASES):
    """P_i = k_i * G with k_i an odd 128-bit integer from the stream."""
    g = SplitMix64(seed + i * STREAM_MUL)
    k = (g.next() | 1) | (g.next() << 64)
    return ec_mul(k, cv.gen, cv)


def synth_bases(n, cv, seed=SEED_BASES):
    return [synth_base(i, cv, seed) for i in range(n)]


# ---------------------------------------------------------------- ProtoGalaxy polynomials
# Restatement of reference src/nifs/protogalaxy/poly/mod.rs (compute_F :66-179, compute_G :218-303,
# compute_K :339-382), src/polynomial/lagrange.rs and folded_trace.rs over bn256::Fr with Python
# integers.  Gates are expression tuples (eval_expression above); a trace is
# dict(challenges=[int], W=[[int]]) with ONE witness vector holding num_advice columns of 2^k rows
# and no lookups, so that advice column j is W[0][j * rows : (j + 1) * rows]
# (src/plonk/eval.rs:171-176).  structure = dict(k, gates, selectors, fixed, num_advice).
def _pg_getter(structure, trace):
    rows = 1 << structure["k"]
    w = trace["W"][0]
    return dict(selectors=structure["selectors"], fixed=structure["fixed"], challenges=trace["challenges"],
                advice=[w[j * rows:(j + 1) * rows] for j in range(structure["num_advice"])])


def pg_iter_evaluate_witness(structure, trace):
    """[gate1(row0), ..., gate1(rowN), gate2(row0), ...]  (src/plonk/mod.rs:1158-1178)"""
    rows, getter = 1 << structure["k"], _pg_getter(structure, trace)
    return [eval_expression(g, getter, r, rows, R_MOD) for g in structure["gates"] for r in range(rows)]


def pg_cyclic_subgroup(log_n):
    w, out, v = get_omega_or_inv(log_n, False), [], 1          # lagrange.rs:22-26
    for _ in range(1 << log_n):
        out.append(v); v = v * w % R_MOD
    return out


def pg_vanish(log_n, point):
    return (pow(point, 1 << log_n, R_MOD) - 1) % R_MOD         # lagrange.rs:81-83


def pg_lagrange(X, log_n):
    n, z, out = 1 << log_n, pg_vanish(log_n, X), []            # lagrange.rs:50-74
    inv_n = pow(n, R_MOD - 2, R_MOD)
    for value in pg_cyclic_subgroup(log_n):
        d = (X - value) % R_MOD
        out.append(1 if (z == 0 and d == 0) else value * inv_n * z * pow(d, R_MOD - 2, R_MOD) % R_MOD)
    return out


def _pg_tree_reduce(nodes, combine):
    """itertools::tree_reduce on a power-of-two number of items: adjacent pairs, level by level
    (any other count reaches `unreachable!` in the reference)"""
    assert len(nodes) & (len(nodes) - 1) == 0 and nodes
    height = 0
    while len(nodes) > 1:
        nodes = [combine(nodes[i], nodes[i + 1], height) for i in range(0, len(nodes), 2)]
        height += 1
    return nodes[0]


def pg_compute_F(betas, delta, structure, trace):
    leaves = pg_iter_evaluate_witness(structure, trace)
    count = len(leaves)
    if count == 0:
        return []
    levels = (count - 1).bit_length()
    points_count = 1
    while points_count < levels:
        points_count *= 2
    log_points = points_count.bit_length() - 1
    betas = list(betas)[:levels]
    powers = [[(b + X * delta) % R_MOD for b in betas] for X in pg_cyclic_subgroup(log_points)]   # :104-113
    # a node holds one value per challenge; a leaf is the same value for all of them (:131-166)
    root = _pg_tree_reduce([[v] * points_count for v in leaves],
                           lambda l, r, h: [(a + b * powers[p][h]) % R_MOD for p, (a, b) in enumerate(zip(l, r))])
    ifft(root, log_points)                                                                   # :169-172
    return root


def pg_fold_traces(points, accumulator, traces):
    """FoldedTrace::new (folded_trace.rs:19-131, 133-179)"""
    all_traces = [accumulator] + list(traces)
    log_n = 0
    while (1 << log_n) < len(traces) + 1:
        log_n += 1
    out = []
    for X in points:
        L = pg_lagrange(X, log_n)
        W = [[sum(l * t["W"][c][i] for l, t in zip(L, all_traces)) % R_MOD for i in range(len(col))] for c, col in enumerate(accumulator["W"])]
        ch = [sum(l * t["challenges"][i] for l, t in zip(L, all_traces)) % R_MOD for i in range(len(accumulator["challenges"]))]
        out.append(dict(challenges=ch, W=W))
    return out


def pg_compute_G(structure, betas_stroke, accumulator, traces, max_degree):
    rows, count = 1 << structure["k"], (1 << structure["k"]) * len(structure["gates"])
    if count == 0:
        return []
    points_count = 1
    while points_count < len(traces) * max_degree + 1:
        points_count *= 2
    log_points = points_count.bit_length() - 1
    levels = (count - 1).bit_length()
    bs = list(betas_stroke)[:levels]
    folded = pg_fold_traces(pg_cyclic_subgroup(log_points), accumulator, traces)
    per_point = [pg_iter_evaluate_witness(structure, ft) for ft in folded]
    leaves = [[per_point[p][i] for p in range(points_count)] for i in range(count)]        # try_multi_product, :258-262
    root = _pg_tree_reduce(leaves, lambda l, r, h: [(a + b * bs[h]) % R_MOD for a, b in zip(l, r)])   # :263-290
    ifft(root, log_points)
    return root


def pg_compute_K(structure, f_alpha, betas_stroke, accumulator, traces, max_degree):
    g = pg_compute_G(structure, betas_stroke, accumulator, traces, max_degree)
    log_n = len(g).bit_length() - 1
    g_evals = list(g)
    coset_fft(g_evals)                                                                       # :358
    k_evals = []
    for w, g_y in zip(pg_cyclic_subgroup(log_n), g_evals):
        pt = FR_ZETA * w % R_MOD
        l_y = f_alpha * pg_lagrange(pt, log_n)[0] % R_MOD
        k_evals.append((g_y - l_y) * pow(pg_vanish(log_n, pt), R_MOD - 2, R_MOD) % R_MOD)  # :360-376
    return coset_ifft(k_evals)
