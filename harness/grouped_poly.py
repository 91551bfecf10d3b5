"""Host-side mirror of `GroupedPoly` (reference src/polynomial/grouped_poly.rs): an expression grouped by
the power of the folding variable, `x^0 a + x^1 b + x^3 c -> [a, b, None, c]`.

`GroupedPoly.new(expr, ctx)` substitutes every advice / lookup query Z_i by Z_i + X * Z_(i + num_fold_vars) and
every challenge r_j by r_j + X * r_(j + num_challenges) (grouped_poly.rs:88-140) and multiplies out symbolically;
terms()[k] is the coefficient of X^k.  `iter_from_first` (:154-156) yields the cross terms T_1 .. T_d that
`commit_cross_terms` evaluates row by row and commits (src/nifs/vanilla/mod.rs:100-127).  The trees are built
node for node like the reference's (its `Display` tests are asserted on them, tests/golden/ref_kats.json).
"""
from itertools import zip_longest

from .expression import ADVICE, LOOKUP, Challenge, Constant, Expression, Negated, Polynomial, Product, Scaled, Sum


class GroupedPoly:
    def __init__(self, terms=None):
        self.terms = list(terms or [])                     # list of Expression | None, index = degree

    @classmethod
    def from_terms(cls, pairs):
        """`impl From<IntoIterator<(usize, Expression)>>`, grouped_poly.rs:46-62 (a dict or (degree, expr) pairs)"""
        self = cls()
        for degree, expr in (pairs.items() if isinstance(pairs, dict) else pairs):
            if degree >= len(self.terms):
                self.terms.extend([None] * (degree + 1 - len(self.terms)))
            self.terms[degree] = expr
        return self

    @classmethod
    def new(cls, expr, ctx):
        """grouped_poly.rs:88-140"""
        if isinstance(expr, Constant):
            return cls([Constant(expr.value)])
        if isinstance(expr, Polynomial):
            terms = [Polynomial(expr.index, expr.rotation)]
            sub = expr.subtype(ctx)
            if sub == ADVICE:
                terms.append(Polynomial(ctx.shift_advice_index(expr.index), expr.rotation))
            elif sub == LOOKUP:
                terms.append(Polynomial(ctx.shift_lookup_index(expr.index), expr.rotation))
            return cls(terms)
        if isinstance(expr, Challenge):
            return cls([Challenge(expr.index), Challenge(expr.index + ctx.num_challenges)])
        if isinstance(expr, Negated):
            return -cls.new(expr.a, ctx)
        if isinstance(expr, Sum):
            a, b = cls.new(expr.a, ctx), cls.new(expr.b, ctx)
            return a + b
        if isinstance(expr, Product):
            a, b = cls.new(expr.a, ctx), cls.new(expr.b, ctx)
            return a * b
        if isinstance(expr, Scaled):
            return cls.new(expr.a, ctx) * expr.factor
        raise TypeError(f"not an Expression: {expr!r}")

    # ---- grouped_poly.rs:142-168
    def iter_with_degree(self):
        return [(d, e) for d, e in enumerate(self.terms) if e is not None]

    def iter(self):
        return list(self.terms)

    def iter_from_first(self):
        return self.terms[1:]

    def __len__(self):
        return len(self.terms)

    def is_empty(self):
        return not self.terms

    def get(self, index):
        return self.terms[index] if index < len(self.terms) else None

    # ---- impl_poly_ops!, grouped_poly.rs:171-201
    def _zip(self, rhs, rhs_expr):
        out = []
        for l, r in zip_longest(self.terms, rhs.terms):
            if l is not None and r is not None:
                out.append(Sum(l, rhs_expr(r)))
            elif r is not None:
                out.append(rhs_expr(r))
            else:
                out.append(l)
        return GroupedPoly(out)

    def __add__(self, rhs):
        return self._zip(rhs, lambda e: e)

    def __sub__(self, rhs):
        return self._zip(rhs, lambda e: Negated(e))

    def __neg__(self):                                     # :272-285
        return GroupedPoly([None if e is None else Negated(e) for e in self.terms])

    def __mul__(self, other):
        if not isinstance(other, GroupedPoly):             # `impl Mul<&F>`, :203-219: Constant(k) * term
            return GroupedPoly([None if e is None else Product(Constant(other), e) for e in self.terms])
        # :221-270 -- the longer operand (the right one on a tie) is the outer loop, both walked from the top degree
        lhs, rhs = (other, self) if len(self.terms) <= len(other.terms) else (self, other)
        res = []
        rhs_terms = [(d, e) for d, e in enumerate(rhs.terms) if e is not None][::-1]
        for ld, le in [(d, e) for d, e in enumerate(lhs.terms) if e is not None][::-1]:
            for rd, re_ in rhs_terms:
                degree = ld + rd
                expr = Product(le, re_)
                if degree >= len(res):
                    res.extend([None] * (degree + 1 - len(res)))
                res[degree] = expr if res[degree] is None else Sum(res[degree], expr)
        return GroupedPoly(res)
