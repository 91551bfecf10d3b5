"""Host-side mirror of the reference's polynomial expressions (src/polynomial/expression.rs) -- the
structures `commit_cross_terms` (src/nifs/vanilla/mod.rs:79-140) gets its cross-term graphs from:

    gates --compress_expression--> one expression --homogeneous--> uniform degree d
          --GroupedPoly::new (grouped_poly.py)--> d + 1 expressions, the coefficients of X^k in
          f(W1 + X W2, c1 + X c2); terms 1 .. d are the cross terms T_k the fold step evaluates and commits.

The transformations build node-identical trees (the `Display` strings of the reference's own tests
are asserted on them, tests/golden/ref_kats.json), so GraphEvaluator.new(...) on a term yields the
calculation graph the reference would evaluate.  Constants are Python integers below the modulus.

    Expression, Query        expression.rs:69-120      QueryIndexContext       :38-67
    to_string (`Display`)    :121-125, 262-301          evaluate                :186-227
    fold_transform           :233-260                   homogeneous             :356-430
    degree                   :432-449                   challenge_in_degree     :501-513
    compress_expression      src/plonk/util.rs:97-117   CompressedGates         src/plonk/mod.rs:79-134
"""
from dataclasses import dataclass, replace


@dataclass
class QueryIndexContext:
    """expression.rs:38-67"""
    num_selectors: int = 0
    num_fixed: int = 0
    num_advice: int = 0
    num_challenges: int = 0
    num_lookups: int = 0

    def num_fold_vars(self):
        return self.num_advice + self.num_lookups * 5

    def shift_advice_index(self, advice_poly_index):
        return advice_poly_index + self.num_fold_vars()

    def shift_lookup_index(self, lookup_poly_index):
        return lookup_poly_index + self.num_fold_vars()

    def copy(self):
        return replace(self)


SELECTOR, FIXED, ADVICE, LOOKUP = "Selector", "Fixed", "Advice", "Lookup"


def _hex(v):
    """trim_leading_zeros(format!("{:?}", field element)) (src/util.rs:160-164): zero prints as `0x`."""
    return "0x" + format(int(v), "x").lstrip("0")


class Expression:
    """expression.rs:112-120.  The operators build the same nodes as the reference's `impl Add / Sub /
    Mul / Neg` (:470-499): a - b = Sum(a, Negated(b)), a * F = Scaled(a, F)."""

    def to_tuple(self):
        """Neutral nested-tuple form (what the test oracle evaluates)."""
        raise NotImplementedError

    def __add__(self, other):
        return Sum(self, other)

    def __sub__(self, other):
        return Sum(self, Negated(other))

    def __mul__(self, other):
        return Product(self, other) if isinstance(other, Expression) else Scaled(self, other)

    def __neg__(self):
        return Negated(self)

    def __eq__(self, other):
        return isinstance(other, Expression) and self.to_tuple() == other.to_tuple()

    def __hash__(self):
        return hash(self.to_tuple())

    def __str__(self):
        return self.visualize()

    __repr__ = __str__

    # ---- expression.rs:186-227
    def evaluate(self, constant, poly, challenge, negated, sum_, product, scaled):
        ev = lambda e: e.evaluate(constant, poly, challenge, negated, sum_, product, scaled)
        if isinstance(self, Constant):
            return constant(self.value)
        if isinstance(self, Polynomial):
            return poly(self)
        if isinstance(self, Challenge):
            return challenge(self.index)
        if isinstance(self, Negated):
            return negated(ev(self.a))
        if isinstance(self, Sum):
            a = ev(self.a)
            return sum_(a, ev(self.b))
        if isinstance(self, Product):
            a = ev(self.a)
            return product(a, ev(self.b))
        if isinstance(self, Scaled):
            return scaled(ev(self.a), self.factor)
        raise TypeError(f"not an Expression: {self!r}")

    def num_challenges(self):
        """expression.rs:160-184: the number of DISTINCT challenge indices"""
        return len(self.evaluate(lambda c: set(), lambda p: set(), lambda i: {i}, lambda a: a, lambda a, b: a | b, lambda a, b: a | b, lambda a, k: a))

    def poly_set(self):
        """expression.rs:128-157, as a sorted list of ("poly", rotation, index) / ("chal", index)"""
        s = self.evaluate(lambda c: set(), lambda p: {("poly", p.rotation, p.index)}, lambda i: {("chal", i)}, lambda a: a,
                          lambda a, b: a | b, lambda a, b: a | b, lambda a, k: a)
        key = lambda t: (t[1], t[2], 0) if t[0] == "poly" else (0, t[1], 1)              # the reference's Ord, :24-36
        return sorted(s, key=key)

    def degree(self, ctx):
        """expression.rs:432-449"""
        return self.evaluate(lambda c: 0, lambda p: 1 if p.subtype(ctx) in (ADVICE, LOOKUP) else 0, lambda i: 1, lambda a: a,
                             lambda a, b: max(a, b), lambda a, b: a + b, lambda a, k: a)

    def fold_transform(self, mm, nn):
        """expression.rs:233-260: P(f_1..f_mm, x_1..x_nn) -> P(f, x + r y), r = Challenge(2 * num_challenges)"""
        num_challenges = self.num_challenges()
        r = Challenge(2 * num_challenges)

        def poly(p):
            if p.index < mm:
                return Polynomial(p.index, p.rotation)
            return Polynomial(p.index, p.rotation) + r * Polynomial(p.index + nn, p.rotation)
        return self.evaluate(lambda c: Constant(c), poly, lambda i: Challenge(i) + r * Challenge(i + num_challenges),
                             lambda a: -a, lambda a, b: a + b, lambda a, b: a * b, lambda a, k: a * k)

    def visualize(self):
        """expression.rs:262-301"""
        if isinstance(self, Constant):
            return _hex(self.value)
        if isinstance(self, Polynomial):
            rot = "" if self.rotation == 0 else f"[{self.rotation}]" if self.rotation < 0 else f"[+{self.rotation}]"
            return f"Z_{self.index}{rot}"
        if isinstance(self, Challenge):
            return f"r_{self.index}"
        if isinstance(self, Negated):
            return f"-{self.a}"
        if isinstance(self, Sum):
            return f"{self.a} - {self.b.a}" if isinstance(self.b, Negated) else f"{self.a} + {self.b}"
        if isinstance(self, Product):
            side = lambda e: f"({e.visualize()})" if isinstance(e, Sum) else e.visualize()
            return f"{side(self.a)} * {side(self.b)}"
        if isinstance(self, Scaled):
            return f"\"{_hex(self.factor)}\" * {self.a}"               # `{:?}` of a String: quoted
        raise TypeError(f"not an Expression: {type(self)}")

    def homogeneous(self, ctx):
        """expression.rs:356-430 -> (expression of uniform degree, that degree); a term of lower degree is
        multiplied by Challenge(ctx.num_challenges) to the missing power."""
        new_challenge_index = ctx.num_challenges
        if isinstance(self, Constant):
            return Constant(self.value), 0
        if isinstance(self, Polynomial):
            return Polynomial(self.index, self.rotation), (1 if self.subtype(ctx) in (ADVICE, LOOKUP) else 0)
        if isinstance(self, Challenge):
            return Challenge(self.index), 1
        if isinstance(self, Negated):
            e, d = self.a.homogeneous(ctx)
            return Negated(e), d
        if isinstance(self, Sum):
            (lhs, ld), (rhs, rd) = self.a.homogeneous(ctx), self.b.homogeneous(ctx)
            if ld > rd:
                return lhs + (rhs * challenge_in_degree(new_challenge_index, ld - rd)), ld
            if ld < rd:
                return (lhs * challenge_in_degree(new_challenge_index, rd - ld)) + rhs, rd
            return lhs + rhs, ld
        if isinstance(self, Product):
            (lhs, ld), (rhs, rd) = self.a.homogeneous(ctx), self.b.homogeneous(ctx)
            return lhs * rhs, ld + rd
        if isinstance(self, Scaled):
            e, d = self.a.homogeneous(ctx)
            return Scaled(e, self.factor), d
        raise TypeError(f"not an Expression: {type(self)}")


class Constant(Expression):
    def __init__(self, value):
        self.value = int(value)

    def to_tuple(self):
        return ("const", self.value)


class Polynomial(Expression):
    """A column query (`Query`, expression.rs:69-74): `index` into selectors | fixed | advice | lookup,
    `rotation` relative to the row."""

    def __init__(self, index, rotation=0):
        self.index, self.rotation = int(index), int(rotation)

    def to_tuple(self):
        return ("poly", self.index, self.rotation)

    def subtype(self, ctx):
        """Query::subtype, expression.rs:83-100"""
        if self.index < ctx.num_selectors:
            return SELECTOR
        if self.index < ctx.num_selectors + ctx.num_fixed:
            return FIXED
        if self.index < ctx.num_selectors + ctx.num_fixed + ctx.num_advice:
            return ADVICE
        if self.index < ctx.num_selectors + ctx.num_fixed + ctx.num_advice + 5 * ctx.num_lookups:
            return LOOKUP
        raise IndexError(f"unknown index {self.index} in {ctx}")       # unreachable!() in the reference


class Challenge(Expression):
    def __init__(self, index):
        self.index = int(index)

    def to_tuple(self):
        return ("chal", self.index)


class Negated(Expression):
    def __init__(self, a):
        self.a = a

    def to_tuple(self):
        return ("neg", self.a.to_tuple())


class Sum(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def to_tuple(self):
        return ("sum", self.a.to_tuple(), self.b.to_tuple())


class Product(Expression):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def to_tuple(self):
        return ("prod", self.a.to_tuple(), self.b.to_tuple())


class Scaled(Expression):
    def __init__(self, a, factor):
        self.a, self.factor = a, int(factor)

    def to_tuple(self):
        return ("scaled", self.a.to_tuple(), self.factor)


def challenge_in_degree(new_challenge_index, degree):
    """expression.rs:501-513: Challenge(i) * Challenge(i) * ... (`degree` factors, left-associated)"""
    result = Challenge(new_challenge_index)
    for _ in range(2, degree + 1):
        result = result * Challenge(new_challenge_index)
    return result


def compress_expression(exprs, challenge_index):
    """src/plonk/util.rs:97-117: random linear combination of the gates with one challenge y,
    fold(0, |acc, e| e + acc * y); a single gate stays as it is."""
    y = Challenge(challenge_index)
    if len(exprs) > 1:
        acc = Constant(0)
        for e in exprs:
            acc = Sum(e, Product(acc, y))
        return acc
    return exprs[0] if exprs else Constant(0)


class CompressedGates:
    """src/plonk/mod.rs:79-134.  `new` mutates ctx.num_challenges as the reference does: first to the
    challenges of the compressed expression, then to those of the homogeneous one."""

    def __init__(self, compressed, homogeneous, degree, grouped):
        self.compressed, self.homogeneous, self.degree, self.grouped = compressed, homogeneous, degree, grouped

    @classmethod
    def new(cls, original_expressions, ctx):
        from .grouped_poly import GroupedPoly
        compressed = compress_expression(original_expressions, ctx.num_challenges)
        ctx.num_challenges = compressed.num_challenges()
        homogeneous, degree = compressed.homogeneous(ctx)
        ctx.num_challenges = homogeneous.num_challenges()
        return cls(compressed, homogeneous, degree, GroupedPoly.new(homogeneous, ctx))
