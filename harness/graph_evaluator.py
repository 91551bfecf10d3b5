"""TEST / BENCH HARNESS, not part of the engine: the symbolic half of the reference's cross-term evaluation, restated in
Python so that the device evaluator (mira_amd/graph_evaluator.py) can be fed the reference's own graphs.

* `GraphEvaluator.new(expr)`: the calculation graph the reference builds from an expression
  (src/polynomial/graph_evaluator.rs:196-352) -- same constants table, rotation table, sharing of repeated
  sub-expressions and simplifications, so graphs are comparable node for node;
* `CrossTermPlan(homogeneous, degree, ctx)`: the d + 1 evaluation-point graphs and the interpolation coefficients of
  mira_amd.graph_evaluator.CrossTermPlan, derived from the homogeneous gate polynomial.

Everything the engine's module exports is re-exported, so `from harness import graph_evaluator as G` serves a test as the
one module did before the split."""
from mira_amd.graph_evaluator import *                                       # noqa: F401,F403
from mira_amd.graph_evaluator import (FIELD_FR, MODULUS, OP_ADD, OP_DOUBLE, OP_MUL, OP_NEGATE, OP_SQUARE, OP_STORE, OP_SUB, SRC_CHALLENGE,
                                      SRC_COLUMN, SRC_CONSTANT, SRC_INTERMEDIATE, to_montgomery)   # noqa: F401
from mira_amd import graph_evaluator as _engine

from .expression import Challenge, Constant, Expression, Negated, Polynomial, Product, Scaled, Sum   # noqa: F401

# value sources are tuples ordered like the reference's derived PartialOrd (graph_evaluator.rs:55-68):
# (SRC_CONSTANT, id) < (SRC_INTERMEDIATE, id); columns and challenges only appear inside Store
_ZERO, _ONE, _TWO = (SRC_CONSTANT, 0), (SRC_CONSTANT, 1), (SRC_CONSTANT, 2)


class GraphEvaluator(_engine.GraphEvaluator):
    def __init__(self, field=FIELD_FR):
        super().__init__(field)                   # constants 0, 1, 2: the defaults of graph_evaluator.rs:183-192
        self._known = {}                          # calculation -> intermediate that already holds it

    @classmethod
    def new(cls, expr, field=FIELD_FR):
        """graph_evaluator.rs:196-203"""
        ge = cls(field)
        ge._calc((OP_STORE, ge._expr(expr)))
        return ge

    def _rotation(self, rot):                                   # add_rotation, :206-219
        if rot not in self.rotations:
            self.rotations.append(rot)
        return self.rotations.index(rot)

    def _constant(self, v):                                     # add_constant, :222-235
        v %= self.mod
        if v not in self.constants:
            self.constants.append(v)
        return (SRC_CONSTANT, self.constants.index(v))

    def _calc(self, calc):                                      # add_calculation, :241-258
        if calc not in self._known:
            self._known[calc] = len(self.calculations)
            self.calculations.append(calc)
        return (SRC_INTERMEDIATE, self._known[calc])

    def _expr(self, e):                                         # add_expression, :261-352
        if isinstance(e, Constant):
            return self._constant(e.value)
        if isinstance(e, Polynomial):
            return self._calc((OP_STORE, (SRC_COLUMN, e.index, self._rotation(e.rotation))))
        if isinstance(e, Challenge):
            return self._calc((OP_STORE, (SRC_CHALLENGE, e.index)))
        if isinstance(e, Negated):
            if isinstance(e.a, Constant):
                return self._constant(-e.a.value)
            a = self._expr(e.a)
            return a if a == _ZERO else self._calc((OP_NEGATE, a))
        if isinstance(e, Sum):
            if isinstance(e.b, Negated):                        # a + (-b) is a subtraction
                a, b = self._expr(e.a), self._expr(e.b.a)
                if a == _ZERO:
                    return self._calc((OP_NEGATE, b))
                return a if b == _ZERO else self._calc((OP_SUB, a, b))
            a, b = self._expr(e.a), self._expr(e.b)
            return self._calc((OP_ADD,) + ((a, b) if a <= b else (b, a)))
        if isinstance(e, Product):
            a, b = self._expr(e.a), self._expr(e.b)
            if _ZERO in (a, b):
                return _ZERO
            if a == _ONE:
                return b
            if b == _ONE:
                return a
            if a == _TWO:
                return self._calc((OP_DOUBLE, b))
            if b == _TWO:
                return self._calc((OP_DOUBLE, a))
            if a == b:
                return self._calc((OP_SQUARE, a))
            return self._calc((OP_MUL,) + ((a, b) if a <= b else (b, a)))
        if isinstance(e, Scaled):
            f = e.factor % self.mod
            if f == 0:
                return _ZERO
            if f == 1:
                return self._expr(e.a)
            c = self._constant(f)
            return self._calc((OP_MUL, self._expr(e.a), c))
        raise TypeError(f"not an Expression: {e!r}")


class CrossTermPlan(_engine.CrossTermPlan):
    """The plan of mira_amd.graph_evaluator.CrossTermPlan for a homogeneous gate polynomial (its docstring has the algebra)."""

    def __init__(self, homogeneous, degree, ctx, field=FIELD_FR):
        mod = MODULUS[field]
        nsf, shift, nc = ctx.num_selectors + ctx.num_fixed, ctx.num_fold_vars(), ctx.num_challenges
        d = degree

        def at(x):
            """f with every folded variable v replaced by v1 + x v2 (x = None: by v2 -- the leading coefficient)"""
            def poly(p):
                if p.index < nsf:
                    return Polynomial(p.index, p.rotation)
                second = Polynomial(p.index + shift, p.rotation)
                if x is None:
                    return second
                return fold(Polynomial(p.index, p.rotation), second)

            def fold(first, second):                                       # first + x * second in the cheapest calculations
                if x == 0:
                    return first
                two = Product(Constant(2), second)                         # (2 * v is a DOUBLE)
                mag = second if abs(x) == 1 else two if abs(x) == 2 else Sum(two, second) if abs(x) == 3 else Scaled(second, abs(x))   # 3 v = 2 v + v: no product
                return Sum(first, mag if x > 0 else Negated(mag))          # a + (-b) is one SUB

            def chal(i):
                return Challenge(i + nc) if x is None else fold(Challenge(i), Challenge(i + nc))
            return homogeneous.evaluate(lambda c: Constant(c), poly, chal, lambda a: Negated(a), lambda a, b: Sum(a, b), lambda a, b: Product(a, b),
                                        lambda a, k: Scaled(a, k))
        xs = [(j // 2 + 1) * (1 if j % 2 == 0 else -1) for j in range(d - 1)]     # 1, -1, 2, -2, 3, ...: the cheapest folds
        points = [0, None] + xs                                            # vector order: p_0, p_inf, p_x ...
        evaluators = [GraphEvaluator.new(at(x), field) for x in points]
        # Ainv over the field: Gauss-Jordan on the (d - 1) x (d - 1) matrix x^k
        m = d - 1
        A = [[pow(x, k, mod) for k in range(1, d)] + [1 if j == r else 0 for j in range(m)] for r, x in enumerate(xs)]
        for col in range(m):
            piv = next(r for r in range(col, m) if A[r][col])
            A[col], A[piv] = A[piv], A[col]
            inv = pow(A[col][col], mod - 2, mod)
            A[col] = [v * inv % mod for v in A[col]]
            for r in range(m):
                if r != col and A[r][col]:
                    f = A[r][col]
                    A[r] = [(a - f * b) % mod for a, b in zip(A[r], A[col])]
        ainv = [row[m:] for row in A]                                      # ainv[k - 1][index of x]
        # coefficients of T_k (k = 1 .. d - 1) on [p_0, p_inf, p_x ...]
        coeffs = [[(-sum(ainv[k])) % mod, (-sum(ainv[k][r] * pow(x, d, mod) for r, x in enumerate(xs))) % mod] + ainv[k] for k in range(m)]
        super().__init__(degree, field, evaluators, coeffs)
        self.points = points

    @classmethod
    def from_compressed_gates(cls, cg, ctx, field=FIELD_FR):
        """cg: expression.CompressedGates (homogeneous form + degree), ctx: the QueryIndexContext after CompressedGates.new"""
        return cls(cg.homogeneous, cg.degree, ctx, field)

