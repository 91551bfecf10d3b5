"""TEST / BENCH HARNESS (not the engine): the lookup argument of the reference's special-soundness protocol and the one
circuit its folding tests run it on.

* `lookup_expressions` -- src/plonk/lookup.rs:133-210: for every lookup argument the vanishing polynomials L - l, T - t and
  the log-derivative relations h (l + r) - 1, g (t + r) - m over the five lookup variables (l, t, m, h, g) that follow the
  advice columns in the index space of `Expression::Polynomial` (index num_selectors + num_fixed + num_advice + 5 i + ...);
* `fibo_lookup_gates` / `compressed_fibo_lookup` -- `FiboCircuitWithLookup` (src/nifs/tests.rs:258-530: an addition gate
  and ONE vector lookup of three expressions into a three-column XOR table), its gates in the order
  `ConstraintSystemMetainfo::build` collects them (src/table/constraint_system_metainfo.rs:34-60) and the compressed /
  homogeneous / grouped forms of `PlonkStructure::custom_gates_lookup_compressed` (challenges r1, r2 of the lookup, r3 for the
  random linear combination of the gates, u for the relaxation);
* `LookupTrace` -- a satisfying trace of that circuit as the three-round protocol leaves it (run_sps_protocol_3,
  src/plonk/mod.rs:841-907): W1 = the advice columns, W2 = (l, t, m) (`evaluate_coefficient_1`, lookup.rs:323-346),
  W3 = (h, g) (`evaluate_coefficient_2`, :357-366), with the challenges handed in (the random oracle is not part of the path).

Plain Python integers below the modulus; `to_device` uploads the columns in the reference's Montgomery layout."""
import numpy as np

from .expression import Challenge, CompressedGates, Constant, Polynomial, Product, QueryIndexContext, Sum

NUM_SELECTORS, NUM_FIXED, NUM_ADVICE, NUM_LOOKUPS = 2, 3, 3, 1       # s_add, s_xor | the XOR table | a, b, out | "xor"
S_ADD, S_XOR = 0, 1


def compress_vector(exprs, challenge_index):
    """compress_halo2_expression (src/plonk/util.rs:73-93): fold(0, |acc, e| e + acc * y); one expression stays as it is"""
    if len(exprs) == 1:
        return exprs[0]
    acc, y = Constant(0), Challenge(challenge_index)
    for e in exprs:
        acc = Sum(e, Product(acc, y))
    return acc


def lookup_expressions(lookup_polys, table_polys, lookup_offset, has_vector_lookup):
    """Arguments::to_expressions (lookup.rs:133-210): vanishing polynomials of every lookup, then of every table, then
    (lhs, rhs) of the log-derivative relation per lookup; r = challenge 1 with a vector lookup, else 0."""
    var = lambda i, k: Polynomial(lookup_offset + 5 * i + k)
    out = [L - var(i, 0) for i, L in enumerate(lookup_polys)] + [T - var(i, 1) for i, T in enumerate(table_polys)]
    r = Challenge(1 if has_vector_lookup else 0)
    for i in range(len(lookup_polys)):
        l, t, m, h, g = (var(i, k) for k in range(5))
        out.append(h * (l + r) - Constant(1))
        out.append(g * (t + r) - m)
    return out


def fibo_lookup_gates():
    """-> (gates, ctx): the gate list of FiboCircuitWithLookup and the QueryIndexContext CompressedGates::new starts from"""
    sel = lambda i: Polynomial(i)
    fixed = lambda c: Polynomial(NUM_SELECTORS + c)
    advice = lambda c: Polynomial(NUM_SELECTORS + NUM_FIXED + c)
    lhs, rhs, out = advice(0), advice(1), advice(2)
    add_gate = sel(S_ADD) * (lhs + rhs - out)                                   # create_gate("add"), src/nifs/tests.rs:327-334
    L = compress_vector([sel(S_XOR) * lhs, sel(S_XOR) * rhs, sel(S_XOR) * out], 0)   # meta.lookup("xor"), :316-326
    T = compress_vector([fixed(0), fixed(1), fixed(2)], 0)
    gates = [add_gate] + lookup_expressions([L], [T], NUM_SELECTORS + NUM_FIXED + NUM_ADVICE, True)
    ctx = QueryIndexContext(num_selectors=NUM_SELECTORS, num_fixed=NUM_FIXED, num_advice=NUM_ADVICE, num_lookups=NUM_LOOKUPS,
                            num_challenges=2)                                   # a vector lookup: r1, r2 before the combining r3
    return gates, ctx, L, T


def compressed_fibo_lookup():
    gates, ctx, L, T = fibo_lookup_gates()
    cg = CompressedGates.new(gates, ctx)
    return cg, ctx, L, T


def get_sequence(a, b, c, num):
    """src/nifs/tests.rs:518-527"""
    seq = [a, b, c] + [0] * (num - 3)
    for i in range(3, num):
        seq[i] = seq[i - 3] + (seq[i - 2] ^ seq[i - 1])
    return seq


def _eval(expr, row, sel, fix, adv, chal, mod):
    ns, nf = len(sel), len(fix)

    def poly(p):
        assert p.rotation == 0
        if p.index < ns:
            return int(sel[p.index][row])
        if p.index < ns + nf:
            return fix[p.index - ns][row]
        return adv[p.index - ns - nf][row]
    return expr.evaluate(lambda c: c % mod, poly, lambda i: chal[i], lambda a: (-a) % mod, lambda a, b: (a + b) % mod, lambda a, b: a * b % mod,
                         lambda a, k: a * k % mod)


class LookupTrace:
    """One satisfying trace of FiboCircuitWithLookup over 2^k rows.

    seq = (a, b, c, num): the circuit as the reference synthesises it (row 0 = a, b, c; then a XOR row and an addition row
    per further element, src/nifs/tests.rs:494-512) -- needs 2 num - 5 <= 2^k rows; seq = None: a synthetic trace of the same
    shape for large tables -- every row is, by a seeded choice, an XOR row over the table's range, an addition row over the
    whole field, or empty (the copy constraints between rows are the permutation argument's business, not this path's);
    the choice is seeded by `structure_seed` (the selectors belong to the circuit), the values by `seed`."""

    def __init__(self, k, mod, challenges, seq=None, seed=0, structure_seed=0x5354):
        import random
        self.k, self.mod, self.rows = k, mod, 1 << k
        rows, rng = self.rows, random.Random(seed)
        kinds = random.Random(structure_seed)                                # which rows carry which selector is the CIRCUIT's: the same for every trace
        r1, r2, r3 = challenges
        self.challenges = [c % mod for c in challenges]
        s_add, s_xor = [False] * rows, [False] * rows
        adv = [[0] * rows for _ in range(NUM_ADVICE)]
        if seq is not None:
            a, b, c, num = seq
            assert 2 * num - 5 <= rows
            adv[0][0], adv[1][0], adv[2][0] = a, b, c
            row = 1
            for _ in range(3, num):
                x = b ^ c
                adv[0][row], adv[1][row], adv[2][row] = b, c, x; s_xor[row] = True; row += 1
                new_c = a + x
                adv[0][row], adv[1][row], adv[2][row] = a, x, new_c; s_add[row] = True; row += 1
                a, b, c = b, c, new_c
        else:
            for row in range(rows):
                kind = kinds.randrange(4)
                if kind == 0:
                    x, y = rng.randrange(5), rng.randrange(5)
                    adv[0][row], adv[1][row], adv[2][row] = x, y, x ^ y; s_xor[row] = True
                elif kind == 1:
                    x, y = rng.randrange(mod), rng.randrange(mod)
                    adv[0][row], adv[1][row], adv[2][row] = x, y, (x + y) % mod; s_add[row] = True
                elif kind == 2:                                            # an unconstrained row: anything
                    adv[0][row], adv[1][row], adv[2][row] = rng.randrange(mod), rng.randrange(1 << 32), rng.randrange(7)
        fix = [[0] * rows for _ in range(NUM_FIXED)]
        idx = 0
        for x in range(5):                                                  # load_table, src/nifs/tests.rs:430-458
            for y in range(5):
                fix[0][idx], fix[1][idx], fix[2][idx] = x, y, x ^ y
                idx += 1
        self.selectors, self.fixed, self.advice = [s_add, s_xor], fix, adv
        _, _, L, T = fibo_lookup_gates()
        # round 2: l = L(advice), t = T(fixed) under r1; m counts every table value's occurrences among l, once (lookup.rs:213-307)
        l = [_eval(L, row, self.selectors, fix, adv, [r1], mod) for row in range(rows)]
        t = [_eval(T, row, self.selectors, fix, adv, [r1], mod) for row in range(rows)]
        counts, seen, m = {}, set(), []
        for v in l:
            counts[v] = counts.get(v, 0) + 1
        for v in t:
            m.append(0 if v in seen else counts.get(v, 0))
            seen.add(v)
        assert all(v in seen for v in l), "the trace looks a value up that the table does not hold"
        # round 3: h = 1 / (l + r2), g = m / (t + r2), zero where the denominator is (lookup.rs:309-321)
        inv = lambda v: pow(v, mod - 2, mod) if v % mod else 0
        h = [inv((v + r2) % mod) for v in l]
        g = [mv * inv((tv + r2) % mod) % mod for mv, tv in zip(m, t)]
        assert (sum(h) - sum(g)) % mod == 0                                   # is_sat_log_derivative, src/plonk/mod.rs:589-621
        self.lookup = dict(l=l, t=t, m=m, h=h, g=g)
        # W1, W2, W3 of run_sps_protocol_3: concatenate_with_padding of the columns (already 2^k long each)
        self.W = [[v for col in adv for v in col], l + t + m, h + g]

    def variables(self):
        """the 3 + 5 virtual advice columns of eval_advice_var's index space: a, b, out, l, t, m, h, g"""
        lk = self.lookup
        return list(self.advice) + [lk["l"], lk["t"], lk["m"], lk["h"], lk["g"]]

    def check_is_sat(self, compressed):
        """PlonkStructure::is_sat's evaluation (src/plonk/mod.rs:447-477): the compressed gate polynomial vanishes on every row"""
        cols = self.variables()
        for row in range(self.rows):
            assert _eval(compressed, row, self.selectors, self.fixed, cols, self.challenges, self.mod) == 0, row


def to_montgomery_columns(columns, mod):
    """[[int] * rows] -> (len, rows, 4) uint64 of v * 2^256 mod p"""
    out = np.zeros((len(columns), len(columns[0]) if columns else 0, 4), dtype=np.uint64)
    R = (1 << 256) % mod
    for c, col in enumerate(columns):
        for r, v in enumerate(col):
            mv = v * R % mod
            out[c, r] = [(mv >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    return out
